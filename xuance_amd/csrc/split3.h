// Exact 3-way bf16 split of a float32 (csrc/ppo_trunk_bx.hip, the optimiser launches' split mirror stores):
//   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round-to-nearest-even (v_cvt_pk_bf16_f32); x == h + m + l exactly for
//   every finite x whose parts stay normal (8 + 8 + 8 significand bits; |m| <= 2^-8 |x|, |l| <= 2^-16 |x|).
// The library is compiled with -ffp-contract=off: the subtractions below stay subtractions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xrl {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {      // x0 -> bits 0..15, x1 -> bits 16..31
    const f32x2_t v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// two elements at once: packed pairs (x0 in the low half) of the three parts
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    unsigned ph, pm, pl;
    split3_pair(x, 0.f, ph, pm, pl);
    h = (unsigned short)ph; m = (unsigned short)pm; l = (unsigned short)pl;
}

// Element index (inside one plane) of W[n][k], n < 256, k < 128, in the two sections of the split fragment image
// (include/xrl_hip.h, xrl_pack_mid_frags16): a lane's 8 elements of a 32x32x16 operand are contiguous (16 bytes), a wave's load 1 KB.
__host__ __device__ inline int xrl_frag16_fwd_index(int n, int k) {
    const int t = n >> 5, qq = k >> 4;
    return ((t * 8 + ((qq + t) & 7)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
}
__host__ __device__ inline int xrl_frag16_bwd_index(int n, int k) {
    const int kt = k >> 5, q = n >> 4;
    return 256 * 128 + ((kt * 16 + ((q + kt) & 15)) * 64 + (k & 31) + 32 * ((n >> 3) & 1)) * 8 + (n & 7);
}

// A split mirror store of the optimiser launches: map value j <= -2 names element -(j + 2) of a three-plane bf16 image
// (planes `plane` elements apart) instead of a float slot.
__device__ __forceinline__ void mirror_store(float* dst, int j, float v, int plane) {
    if (j >= 0) dst[j] = v;
    else if (j <= -2) {
        unsigned short h, m, l;
        split3(v, h, m, l);
        unsigned short* d = reinterpret_cast<unsigned short*>(dst) + (-(j + 2));
        d[0] = h; d[plane] = m; d[2 * (size_t)plane] = l;
    }
}

}  // namespace xrl
