// Diagnostics (not on the product path): shader-clock / MFMA-issue probes used by tools/microbench_*.py.
#include "common.h"

namespace xrl {
typedef float f32x16 __attribute__((ext_vector_type(16)));

// out[0] = shader cycles (s_memtime), out[1] = constant-rate wall clock ticks, for `iters` dependent 32x32x2 MFMAs
__global__ void __launch_bounds__(64) mfma_chain_kernel(int iters, long long* out, float* sink) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    sink[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[7];
}
}  // namespace xrl

extern "C" int xrl_debug_mfma_chain(int iters, int blocks, long long* out, float* sink, xrl_stream_t stream) {
    hipLaunchKernelGGL(xrl::mfma_chain_kernel, dim3(blocks), dim3(64), 0, xrl::as_stream(stream), iters, out, sink);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
