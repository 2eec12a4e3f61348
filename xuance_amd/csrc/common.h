// Shared helpers for libxrl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/xrl_hip.h"

namespace xrl {

void set_error(const char* fmt, ...);

#define XRL_CHECK_ARG(cond)                                                          \
    do {                                                                             \
        if (!(cond)) {                                                               \
            xrl::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return XRL_EINVAL;                                                       \
        }                                                                            \
    } while (0)

#define XRL_CHECK_HIP(expr)                                                                          \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            xrl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));    \
            return XRL_EHIP;                                                                         \
        }                                                                                            \
    } while (0)

#define XRL_CHECK_LAUNCH() XRL_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(xrl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// compute units of the current device (cached: hipGetDeviceProperties is far too slow for a per-launch query); 0 on error
int device_cu_count();

constexpr int WAVE = 64;  // CDNA4 wavefront

// ---- wavefront / block reductions (64-lane shuffles, no LDS for the wave part) ----
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over a block of up to 1024 threads; result valid in every thread. `scratch` holds >= 16 T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T t = (lane < nw) ? scratch[lane] : T(0);
    t = wave_sum(t);
    return t;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_waitcnt vmcnt(0) + s_barrier on gfx9, i.e.
// it also waits for every global load still in flight -- including the weight prefetch stream the fused kernels issue at
// their start precisely so that it overlaps the phases in between (measured: the first barrier of the rollout step
// stalled ~5 k cycles on the 64 KB B-fragment stream).  Global loads stay pending across this barrier; the compiler still
// inserts the vmcnt wait in front of the first use of each loaded register.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pull every 64-byte line of a by-value kernel argument struct into the scalar cache with ONE round trip.  hipcc loads
// kernel arguments lazily, block by block (s_load ... s_waitcnt, seven times in a row at the top of the rollout kernel);
// right after a kernel boundary each of those is a miss all the way to memory.  After this, they hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
    typedef const __attribute__((address_space(4))) unsigned int* kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NL = (BYTES + 63) / 64;
    static_assert(NL <= 16, "kernel argument struct larger than 1 KB");
    unsigned int t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = ka[(i < NL ? i : NL - 1) * 16];
    asm volatile("" ::"s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]), "s"(t[4]), "s"(t[5]), "s"(t[6]), "s"(t[7]), "s"(t[8]),
                 "s"(t[9]), "s"(t[10]), "s"(t[11]), "s"(t[12]), "s"(t[13]), "s"(t[14]), "s"(t[15]));
}

// The instruction cache is invalidated at every kernel launch and a miss that also misses L2 costs ~4 k cycles (measured:
// a wave that merely branches into a cold code region of the rollout kernel stalls that long).  Sequential fetch is
// prefetched by the hardware, but every far branch target is a fresh miss.  One wave therefore pulls the kernel's own
// code into L2 with ordinary data loads at kernel start (one 128-byte line per lane and load), so that later
// instruction-cache misses become L2 hits.  `KB` must not exceed the kernel's code size.  Returns a value the caller
// keeps alive (and passes to code_prefetch_done) so the loads are not dead.
template <int KB>
__device__ __forceinline__ unsigned code_prefetch_start() {
    const unsigned long long pc = (unsigned long long)__builtin_amdgcn_s_getpc() & ~127ull;
    const unsigned* base = reinterpret_cast<const unsigned*>(pc) + (threadIdx.x & 63) * 32;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < KB / 8; ++k) acc ^= base[k * 2048];          // 64 lanes x 128 B = 8 KB per load
    return acc;
}
__device__ __forceinline__ void code_prefetch_done(unsigned v) { asm volatile("" ::"v"(v)); }

// Compile-time activation variants.  IMPORTANT (measured, tools/microbench_rollout.py with -DXRL_TILE_PROBE): a run-time
// `switch (act)` inside a per-element loop is if-converted by hipcc -- tanhf AND expf are evaluated for every element and
// the result selected -- which cost ~775 cycles per element (12 k cycles per 16-value MFMA epilogue).  Hot loops therefore
// dispatch ONCE on the activation (XRL_ACT_DISPATCH) and run a loop specialised for it.
template <int ACT>
__device__ __forceinline__ float act_apply_c(float z) {
    if (ACT == XRL_ACT_RELU) return z > 0.f ? z : 0.f;
    if (ACT == XRL_ACT_LEAKY_RELU) return fmaxf(z, z * 0.01f);          // the same number as z > 0 ? z : 0.01 z, without the compare -> VCC -> select chain
    if (ACT == XRL_ACT_TANH) return tanhf(z);
    if (ACT == XRL_ACT_SIGMOID) return 1.f / (1.f + expf(-z));
    return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad_c(float y) {
    if (ACT == XRL_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (ACT == XRL_ACT_LEAKY_RELU) return y > 0.f ? 1.f : 0.01f;
    if (ACT == XRL_ACT_TANH) return 1.f - y * y;
    if (ACT == XRL_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}
#define XRL_ACT_DISPATCH(act_value, ...)                                                            \
    switch (act_value) {                                                                            \
        case XRL_ACT_RELU: { constexpr int ACT = XRL_ACT_RELU; __VA_ARGS__ } break;                 \
        case XRL_ACT_LEAKY_RELU: { constexpr int ACT = XRL_ACT_LEAKY_RELU; __VA_ARGS__ } break;     \
        case XRL_ACT_TANH: { constexpr int ACT = XRL_ACT_TANH; __VA_ARGS__ } break;                 \
        case XRL_ACT_SIGMOID: { constexpr int ACT = XRL_ACT_SIGMOID; __VA_ARGS__ } break;           \
        default: { constexpr int ACT = XRL_ACT_NONE; __VA_ARGS__ } break;                           \
    }

__device__ __forceinline__ float act_apply(float z, int act) {
    switch (act) {
        case XRL_ACT_RELU: return z > 0.f ? z : 0.f;
        case XRL_ACT_LEAKY_RELU: return z > 0.f ? z : z * 0.01f;
        case XRL_ACT_TANH: return tanhf(z);
        case XRL_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
        default: return z;
    }
}

// derivative of the activation expressed with its OUTPUT y (what autograd's backward kernels use)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case XRL_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case XRL_ACT_LEAKY_RELU: return y > 0.f ? 1.f : 0.01f;
        case XRL_ACT_TANH: return 1.f - y * y;
        case XRL_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

}  // namespace xrl
