// The optimiser step of the PREVIOUS minibatch as the prologue of the next minibatch launch (xrl_ppo_trunk_chained,
// include/xrl_hip.h): what reduce_adam_kernel<false, 1> (csrc/optim.hip) does in a launch of its own -- the same statements on
// the same elements in the same order, so parameters, moments, clipped gradient, partial sums and the optimiser state are
// bit-identical -- done by the first ceil(P / 256) workgroups of a launch whose workgroups are all resident, followed by the
// hand-over of the new parameters (and their mirror copies) to every workgroup of the launch.
// Replaces, per minibatch, a kernel boundary + argument fetch + cold first round trip of the optimiser launch (reference:
// clip_grad_norm_ + Adam.step + LinearLR.step, ppo_learner.py:61-67).
//
// Two in-launch barriers:
//   B1 (flags sync[4 + b] = step): every workgroup has read the optimiser state; the workers' partial sums of squares are
//      published (relaxed agent-scope atomics, as in reduce_adam_kernel: no payload besides them);
//   B2 (flags sync[4 + XRL_CHAIN_MAX_WGS + b] = step): the workers' parameter / mirror stores go out write-through (agent-scope
//      stores) -> every storing wave drains (vmcnt(0)) -> __syncthreads -> lane 0: relaxed flag store; every workgroup polls the
//      workers' flags with one wave (relaxed), then ONE agent-scope acquire, __syncthreads, plain loads
//      (MI355X guide, Guideline 16 form R1: placement-independent; the per-XCD L2s are not coherent with each other).
// The epoch of both flag sets is the optimiser step this launch performs (st->step + 1): a value no earlier launch has written,
// counted on the device, so a replayed graph needs no per-launch argument.
#pragma once
#include "common.h"

namespace xrl {

constexpr int CHAIN_SCRATCH_FLOATS = 2048 + 256 + 8 + 8;     // gsum [4][64][4] doubles | gtot [256] | 4 doubles | flags

__device__ __forceinline__ double chain_group_sum(double v, double* scratch4, int tid) {
    // group_sum of csrc/optim.hip over threads 0..255 of a 512-thread workgroup (threads >= 256 pass 0 and write nothing)
    v = wave_sum(v);
    const int lane = tid & 63, w = tid >> 6;
    __syncthreads();
    if (lane == 0 && w < 4) scratch4[w] = v;
    __syncthreads();
    double t = (lane < 4) ? scratch4[lane] : 0.0;
    return wave_sum(t);
}

// Every thread of every workgroup of the launch calls this (blockDim.x == 512).  `scratch`: CHAIN_SCRATCH_FLOATS floats of LDS,
// 16-byte aligned, free for the duration of the call.
// cdbg: NULL, or [8 x gridDim.x] real-time-counter stamps (100 MHz) of thread 0 of every workgroup: start | slabs summed | B1 flag out |
// past B1 | Adam stored | released | past B2 poll | acquired   (tools/probe_chain.py)
__device__ __forceinline__ void chain_prologue(const xrl_opt_chain_t& o, float* scratch, long long* cdbg = nullptr) {
#define CSTAMP(k) do { if (cdbg && threadIdx.x == 0) cdbg[8 * blockIdx.x + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    CSTAMP(0);
    typedef double gsum_t[64][4];
    gsum_t* gsum = reinterpret_cast<gsum_t*>(scratch);                               // [4][64][4]
    float* gtot = scratch + 2048;                                                    // [256]
    double* gscratch = reinterpret_cast<double*>(scratch + 2048 + 256);             // [4]
    int* s_flag = reinterpret_cast<int*>(scratch + 2048 + 256 + 8);                 // [0] a wait of this workgroup failed
    const int tid = threadIdx.x, tg = tid & 255;
    const int64_t P = o.P, P4 = P / 4, st4 = o.slab_stride / 4;
    const int n_vb = (int)((P4 + 63) / 64);
    const int vb = blockIdx.x;
    const bool wblock = vb < n_vb;                      // this workgroup owns virtual block vb of reduce_adam_kernel ...
    const bool worker = wblock && tid < 256;                // ... with its first 256 threads
    xrl_adam_state_t* st = o.state;
    unsigned* sync = o.sync;
    const xrl_mirrors_t& mir = o.mirrors;
    const unsigned failed_before = __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) s_flag[0] = 0;
    const int pq = tg & 63, sg = tg >> 6;
    const int64_t qi = (int64_t)vb * 64 + pq;
    int n_split = o.n_split;
    // ---- phase 1: reduce_adam_kernel's vector path for this workgroup's 64 quads
    double sq = 0.0;
    {
        double gx = 0.0, gy = 0.0, gz = 0.0, gw = 0.0;
        if (worker && qi < P4) {
            const float4* src = reinterpret_cast<const float4*>(o.slabs) + qi;
            int s = sg;
            for (; s + 124 < n_split; s += 128) {
                float4 w[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) w[j] = src[(int64_t)(s + 4 * j) * st4];
#pragma unroll
                for (int j = 0; j < 32; ++j) { gx += w[j].x; gy += w[j].y; gz += w[j].z; gw += w[j].w; }
            }
            for (; s + 28 < n_split; s += 32) {
                float4 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = src[(int64_t)(s + 4 * j) * st4];
#pragma unroll
                for (int j = 0; j < 8; ++j) { gx += w[j].x; gy += w[j].y; gz += w[j].z; gw += w[j].w; }
            }
            for (; s < n_split; s += 4) { const float4 w = src[(int64_t)s * st4]; gx += w.x; gy += w.y; gz += w.z; gw += w.w; }
            if (qi * 4 < mir.fold_len) {
                const float4* src2 = src + mir.fold_off / 4;
                s = sg;
                for (; s + 124 < n_split; s += 128) {
                    float4 w[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) w[j] = src2[(int64_t)(s + 4 * j) * st4];
#pragma unroll
                    for (int j = 0; j < 32; ++j) { gx += w[j].x; gy += w[j].y; gz += w[j].z; gw += w[j].w; }
                }
                for (; s + 28 < n_split; s += 32) {
                    float4 w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = src2[(int64_t)(s + 4 * j) * st4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { gx += w[j].x; gy += w[j].y; gz += w[j].z; gw += w[j].w; }
                }
                for (; s < n_split; s += 4) { const float4 w = src2[(int64_t)s * st4]; gx += w.x; gy += w.y; gz += w.z; gw += w.w; }
            }
        }
        if (worker) { gsum[sg][pq][0] = gx; gsum[sg][pq][1] = gy; gsum[sg][pq][2] = gz; gsum[sg][pq][3] = gw; }
        __syncthreads();
        if (worker && sg == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi < P4) {
                double t0 = gsum[0][pq][0], t1 = gsum[0][pq][1], t2 = gsum[0][pq][2], t3 = gsum[0][pq][3];
#pragma unroll
                for (int k = 1; k < 4; ++k) { t0 += gsum[k][pq][0]; t1 += gsum[k][pq][1]; t2 += gsum[k][pq][2]; t3 += gsum[k][pq][3]; }
                t = make_float4((float)t0, (float)t1, (float)t2, (float)t3);
                sq += (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
            }
            *reinterpret_cast<float4*>(&gtot[pq * 4]) = t;
        }
    }
    const double tsum = chain_group_sum(sq, gscratch, tid);
    CSTAMP(1);
    // optimiser scalars (EVERY workgroup reads the state before it publishes its B1 flag: workgroup 0 advances it behind B1)
    const int step = st->step + 1;
    const int k = st->sched_steps < st->total_iters ? st->sched_steps : st->total_iters;
    const double lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k / (double)st->total_iters);
    const double b1 = st->beta1, b2 = st->beta2;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)st->eps;
    const float w1 = (float)(1.0 - b1), fb2 = (float)b2, w2 = (float)(1.0 - b2), wd = (float)st->weight_decay;
    const int64_t i = (int64_t)vb * 256 + tg;
    float p0 = 0.f, m0 = 0.f, v0 = 0.f;
    int mj[XRL_MAX_MIRRORS];
#pragma unroll
    for (int q = 0; q < XRL_MAX_MIRRORS; ++q) mj[q] = -1;
    if (worker && i < P) {
        p0 = o.params[i]; m0 = o.m[i]; v0 = o.v[i];
#pragma unroll
        for (int q = 0; q < XRL_MAX_MIRRORS; ++q) if (q < mir.n) mj[q] = mir.map[q][i];
    }
    // ---- B1: partial sums of squares + "state read" flags
    if (tid == 0 && wblock) __hip_atomic_store(&o.sumsq_part[vb], tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&sync[4 + blockIdx.x], (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CSTAMP(2);
    if (wblock) {                                       // (uniform per workgroup)
        int spins = 0;
        for (;;) {
            int ok = 1;
            for (int j = tid; j < (int)gridDim.x; j += blockDim.x)
                ok &= __hip_atomic_load(&sync[4 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)step;
            if (__syncthreads_and(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { if (tid == 0) { s_flag[0] = 1; __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } break; }
        }
        __syncthreads();
        CSTAMP(3);
        // ---- phase 2: adam_step_kernel for parameter i
        double ssum = 0.0;
        if (tid < 256)
            for (int j = tg; j < o.n_part; j += 256)
                ssum += j < n_vb ? __hip_atomic_load(&o.sumsq_part[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        const double gs = chain_group_sum(ssum, gscratch, tid);
        double total_norm = s_flag[0] ? __builtin_nan("") : sqrt(gs);
        float coef = 1.f;
        if (o.max_norm > 0.0) {
            const double c = o.max_norm / (total_norm + 1e-6);
            coef = (float)(c < 1.0 ? c : 1.0);
        }
        const bool poisoned = s_flag[0] != 0 || total_norm != total_norm || failed_before != 0u;
        if (worker && i < P && !poisoned) {
            float g = gtot[tg] * coef;
            o.grad[i] = g;
            if (wd != 0.f) g += wd * p0;
            const float mi = m0 + (g - m0) * w1;
            const float vi = v0 * fb2 + w2 * g * g;
            o.m[i] = mi; o.v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            const float pn = p0 - step_size * (mi / denom);
            // what the other workgroups read behind B2 goes out WRITE-THROUGH (agent-scope stores = `sc1`): no release fence -- a
            // `buffer_wbl2` per workgroup has every XCD's L2 scanned 17 times in a row (measured: flags visible 4 us after the last one
            // had been stored, 1.4 us for the fence itself) -- just the drain below (guide, Guideline 16 form R1)
            __hip_atomic_store(&o.params[i], pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < XRL_MAX_MIRRORS; ++q)
                if (q < mir.n && mj[q] >= 0) __hip_atomic_store(&mir.dst[q][mj[q]], pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x == 0 && tid == 0) {              // (every workgroup read the state before its B1 flag)
            st->last_grad_norm = total_norm;
            st->step = step;
            const int ns = st->sched_steps + 1;
            st->sched_steps = ns;
            const int k2 = ns < st->total_iters ? ns : st->total_iters;
            st->last_lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k2 / (double)st->total_iters);
        }
        // ---- B2, producer side: this workgroup's stores released at agent scope, then its flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CSTAMP(4);
        if (tid == 0) {
            __hip_atomic_store(&sync[4 + XRL_CHAIN_MAX_WGS + blockIdx.x], (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        CSTAMP(5);
    }
    // ---- B2, consumer side (every workgroup): one wave polls the workers' flags, one acquire, then plain loads
    if (tid < 64) {
        int spins = 0;
        for (;;) {
            int ok = 1;
            for (int j = tid; j < n_vb; j += 64)
                ok &= __hip_atomic_load(&sync[4 + XRL_CHAIN_MAX_WGS + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)step;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000) { if (tid == 0) __hip_atomic_store(&sync[2], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        CSTAMP(6);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    CSTAMP(7);
#undef CSTAMP
}

}  // namespace xrl
