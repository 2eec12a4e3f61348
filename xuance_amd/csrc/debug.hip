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

// Cold vs warm instruction fetch: a straight-line chain of `4 * 512` VOP3 fma (16 KB of code) executed `passes` times by
// every wave; out[pass] = shader cycles of that pass (wave 0 of block 0).  Pass 0 runs on a cold instruction cache.
namespace xrl {
__global__ void __launch_bounds__(512) icache_probe_kernel(int passes, long long* out, float* sink, float a, float b) {
    float x = a + threadIdx.x;
    for (int p = 0; p < passes; ++p) {
        const long long c0 = clock64();
#pragma unroll
        for (int i = 0; i < 512; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %1"
                         : "+v"(x) : "v"(a), "v"(b));
        }
        const long long c1 = clock64();
        if (threadIdx.x == 0 && blockIdx.x == 0) out[p] = c1 - c0;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
}  // namespace xrl

extern "C" int xrl_debug_icache(int passes, int blocks, int threads, long long* out, float* sink, xrl_stream_t stream) {
    hipLaunchKernelGGL(xrl::icache_probe_kernel, dim3(blocks), dim3(threads), 0, xrl::as_stream(stream), passes, out, sink, 1.0001f, 0.5f);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

// Far-branch cost: 16 taken s_branch hops over 2 KB of never-executed padding each; out[pass] = cycles of the 16 hops.
namespace xrl {
__global__ void __launch_bounds__(512) ijump_probe_kernel(int passes, long long* out, float* sink, float a) {
    float x = a + threadIdx.x;
    for (int p = 0; p < passes; ++p) {
        const long long c0 = clock64();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %1\n\ts_branch 1f\n\t.fill 512, 4, 0xbf800000\n1:" : "+v"(x) : "v"(a));
        }
        const long long c1 = clock64();
        if (threadIdx.x == 0 && blockIdx.x == 0) out[p] = c1 - c0;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
}  // namespace xrl

extern "C" int xrl_debug_ijump(int passes, int blocks, int threads, long long* out, float* sink, xrl_stream_t stream) {
    hipLaunchKernelGGL(xrl::ijump_probe_kernel, dim3(blocks), dim3(threads), 0, xrl::as_stream(stream), passes, out, sink, 1.0001f);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
