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

// Compile-time activation variants.  IMPORTANT (measured, tools/microbench_rollout.py with -DXRL_TILE_PROBE): a run-time
// `switch (act)` inside a per-element loop is if-converted by hipcc -- tanhf AND expf are evaluated for every element and
// the result selected -- which cost ~775 cycles per element (12 k cycles per 16-value MFMA epilogue).  Hot loops therefore
// dispatch ONCE on the activation (XRL_ACT_DISPATCH) and run a loop specialised for it.
template <int ACT>
__device__ __forceinline__ float act_apply_c(float z) {
    if (ACT == XRL_ACT_RELU) return z > 0.f ? z : 0.f;
    if (ACT == XRL_ACT_LEAKY_RELU) return z > 0.f ? z : z * 0.01f;
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
