// Diagnostics (NOT part of libxrl_hip.so): shader-clock / MFMA-issue / I-cache / XCD-barrier probes, built by
// tools/probe_lib.py into tools/lib/libxrl_probe.so and used by tools/microbench_{clock,icache,xcd_barrier}.py.
#include "common.h"
#include <cstdarg>
#include <cstdio>

namespace xrl {   // the two helpers common.h expects from the product library
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int device_cu_count() { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }
}  // namespace xrl

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

extern "C" int xrl_probe_mfma_chain(int iters, int blocks, long long* out, float* sink, xrl_stream_t stream) {
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

extern "C" int xrl_probe_icache(int passes, int blocks, int threads, long long* out, float* sink, xrl_stream_t stream) {
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

extern "C" int xrl_probe_ijump(int passes, int blocks, int threads, long long* out, float* sink, xrl_stream_t stream) {
    hipLaunchKernelGGL(xrl::ijump_probe_kernel, dim3(blocks), dim3(threads), 0, xrl::as_stream(stream), passes, out, sink, 1.0001f);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

// Intra-XCD grid barrier probe: workgroups are dealt round-robin to the 8 XCDs, so of a grid of 8*n_wg only those with
// blockIdx % 8 == 0 stay (the rest exit): the survivors share ONE L2.  Each iteration every survivor publishes a value,
// crosses a counter barrier (L2 atomics) and reads every other survivor's value with device-scope loads.
// out[0] = cycles per iteration (workgroup 0), out[1] = number of wrong values seen, out[2] = timeouts, out[3..] = XCC ids.
namespace xrl {
template <int VARIANT>
__device__ __forceinline__ void probe_barrier(unsigned* counter, unsigned want, int* s_to) {
    if (VARIANT == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) { if (++spins > 2000000) { *s_to = 1; break; } }
    } else if (VARIANT == 1) {         // L2-local atomics (workgroup-scope encoding: no fabric round trip), relaxed polling
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        while (__hip_atomic_fetch_add(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { *s_to = 1; break; }
        }
    } else if (VARIANT == 3) {         // formally correct form: release fence, relaxed add + relaxed polling, acquire fence
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { *s_to = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {                           // agent-scope relaxed add, relaxed agent-scope load polling with back-off
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { *s_to = 1; break; }
        }
    }
}

template <int VARIANT>
__global__ void __launch_bounds__(256) xcd_barrier_probe_kernel(int iters, int n_wg, unsigned* counter, float* slots,
                                                                long long* out) {
    if (blockIdx.x % 8 != 0) return;
    const int wg = blockIdx.x / 8;
    __shared__ int s_bad, s_to;
    if (threadIdx.x == 0) { s_bad = 0; s_to = 0; }
    __syncthreads();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[3 + wg] = xcc & 0xf;
    const long long c0 = clock64();
    for (int it = 1; it <= iters; ++it) {
        if (threadIdx.x < 64) slots[wg * 64 + threadIdx.x] = (float)(it * 1000 + wg);
        if (VARIANT == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // by the storing wave itself
        __builtin_amdgcn_s_waitcnt(0);                       // stores acknowledged by L2
        __syncthreads();
        if (threadIdx.x == 0) probe_barrier<VARIANT>(counter, (unsigned)(n_wg * it), &s_to);
        __syncthreads();
        if (s_to) break;
        // read everybody's slot with device-scope loads (bypass this CU's L1)
        if (VARIANT == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // every wave: invalidate this CU's view
        for (int i = threadIdx.x; i < n_wg * 64; i += blockDim.x) {
            const float v = VARIANT == 3 ? slots[i] : __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float)(it * 1000 + i / 64)) atomicAdd(&s_bad, 1);
        }
        __syncthreads();
        // second barrier so nobody overwrites a slot that is still being read
        if (threadIdx.x == 0) probe_barrier<VARIANT>(counter + 32, (unsigned)(n_wg * it), &s_to);
        __syncthreads();
        if (s_to) break;
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) {
        if (wg == 0) out[0] = (c1 - c0) / (iters > 0 ? iters : 1);
        atomicAdd((unsigned long long*)&out[1], (unsigned long long)s_bad);
        atomicAdd((unsigned long long*)&out[2], (unsigned long long)s_to);
    }
}
}  // namespace xrl

extern "C" int xrl_probe_xcd_barrier(int iters, int n_wg, unsigned* counter, float* slots, long long* out, xrl_stream_t stream) {
    XRL_CHECK_ARG(n_wg >= 1 && n_wg <= 32 && counter && slots && out);
    const int variant = iters >> 24;
    iters &= 0xffffff;
    if (variant == 1) hipLaunchKernelGGL(xrl::xcd_barrier_probe_kernel<1>, dim3(8 * n_wg), dim3(256), 0, xrl::as_stream(stream), iters, n_wg, counter, slots, out);
    else if (variant == 3) hipLaunchKernelGGL(xrl::xcd_barrier_probe_kernel<3>, dim3(8 * n_wg), dim3(256), 0, xrl::as_stream(stream), iters, n_wg, counter, slots, out);
    else if (variant == 2) hipLaunchKernelGGL(xrl::xcd_barrier_probe_kernel<2>, dim3(8 * n_wg), dim3(256), 0, xrl::as_stream(stream), iters, n_wg, counter, slots, out);
    else hipLaunchKernelGGL(xrl::xcd_barrier_probe_kernel<0>, dim3(8 * n_wg), dim3(256), 0, xrl::as_stream(stream), iters, n_wg, counter,
                       slots, out);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

// ---- co-issue probe (round 6): what a wave can do NEXT TO a wave of the same SIMD that issues back-to-back fp32 MFMAs.
// One 512-thread workgroup per CU: waves w and w + 4 share a SIMD.  Waves 0..3 run role_a, waves 4..7 role_b:
//   0 idle | 1 chained v_mfma_f32_32x32x2_f32 (one accumulator) | 2 four independent accumulators | 3 VALU: 8 independent fma chains
//   4 LDS reads (ds_read_b128 + one add) | 5 bf16 v_mfma_f32_32x32x16_bf16 chained | 6 VALU transcendental (v_exp_f32 chains)
// prio_b: s_setprio of the role_b waves.  out[wave] = shader cycles of the wave's own loop (iters iterations).
namespace xrl {
typedef float cf32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 cbf16x8;

template <int ROLE>
__device__ __forceinline__ float coissue_role(int iters, float seed, const float* lds) {
    float r = 0.f;
    if (ROLE == 1) {
        cf32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc, 0, 0, 0);
        }
        r = acc[0] + acc[15];
    } else if (ROLE == 2) {
        cf32x16 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = seed;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc[a], 0, 0, 0);
        }
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (ROLE == 3) {
        float c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_fmaf(c[i], 0.999f, seed);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += c[i];
    } else if (ROLE == 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* p = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float4 v = p[64 * ((k + it) & 15)]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        }
        r = s.x + s.y + s.z + s.w;
    } else if (ROLE == 5) {
        cf32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = seed;
        cbf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed; b[i] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        r = acc[0] + acc[15];
    } else if (ROLE == 6) {
        float c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = seed + 0.01f * i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_exp2f(c[i] * 0.25f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += c[i];
    } else if (ROLE >= 7 && ROLE <= 10) {
        // the SAME wave: one f32 MFMA (four accumulators in turn), then NV independent fmas -- 16 MFMAs + 16 NV fmas per iteration
        constexpr int NV = ROLE == 7 ? 4 : ROLE == 8 ? 8 : ROLE == 9 ? 12 : 16;
        cf32x16 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = seed;
        float c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc[k & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NV; ++i) c[i] = __builtin_fmaf(c[i], 0.999f, seed);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += c[i];
        r += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
    return r;
}

__device__ __forceinline__ float coissue_dispatch(int role, int iters, float seed, const float* lds) {
    switch (role) {
        case 1: return coissue_role<1>(iters, seed, lds);
        case 2: return coissue_role<2>(iters, seed, lds);
        case 3: return coissue_role<3>(iters, seed, lds);
        case 4: return coissue_role<4>(iters, seed, lds);
        case 5: return coissue_role<5>(iters, seed, lds);
        case 6: return coissue_role<6>(iters, seed, lds);
        case 7: return coissue_role<7>(iters, seed, lds);
        case 8: return coissue_role<8>(iters, seed, lds);
        case 9: return coissue_role<9>(iters, seed, lds);
        case 10: return coissue_role<10>(iters, seed, lds);
        default: return 0.f;
    }
}

__global__ void __launch_bounds__(512) coissue_probe_kernel(int role_a, int role_b, int iters_a, int iters_b, int prio_b, long long* out, float* sink, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 16 * 4];
    for (int i = threadIdx.x; i < 64 * 16 * 4; i += 512) lds[i] = seed * i;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool b_side = wave >= 4;
    if (b_side && prio_b) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_barrier();
    const long long c0 = clock64();
    const float r = b_side ? coissue_dispatch(role_b, iters_b, seed, lds) : coissue_dispatch(role_a, iters_a, seed, lds);
    asm volatile("" ::"v"(r));
    const long long c1 = clock64();
    if (b_side && prio_b) __builtin_amdgcn_s_setprio(0);
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = c1 - c0;
    if (r == 12345.678f) sink[threadIdx.x] = r;
}
}  // namespace xrl

extern "C" int xrl_probe_coissue(int role_a, int role_b, int iters_a, int iters_b, int prio_b, int blocks, long long* out, float* sink, xrl_stream_t stream) {
    XRL_CHECK_ARG(out && sink && blocks >= 1);
    hipLaunchKernelGGL(xrl::coissue_probe_kernel, dim3(blocks), dim3(512), 0, xrl::as_stream(stream), role_a, role_b, iters_a, iters_b, prio_b, out, sink, 1.0f);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
