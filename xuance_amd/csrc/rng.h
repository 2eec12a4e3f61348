// Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// key = 64-bit seed, counter = (c0, c1, c2, 0): every (env, step, stream) triple has its own reproducible draw, so
// results do not depend on launch geometry or on hipGraph replay.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace xrl {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ void philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t (&out)[4]) {
    uint32_t c[4] = {c0, c1, c2, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int i = 0; i < 10; ++i) philox_round(c, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }          // [0,1)
__device__ __forceinline__ double u01d(uint32_t a, uint32_t b) {
    return (double)((((uint64_t)a << 21) ^ (uint64_t)b) & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}

constexpr uint32_t STREAM_ACTION = 0x41435431u, STREAM_GAUSS = 0x47415500u, STREAM_RESET_A = 0x52455345u,
                   STREAM_RESET_B = 0x52455346u, STREAM_EGREEDY = 0x45475200u;

}  // namespace xrl
