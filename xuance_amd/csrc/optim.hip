// Optimiser kernels: deterministic reduction of the per-chunk gradient slabs, global-norm clipping, Adam and the
// LinearLR schedule.  Replaces  clip_grad_norm_ + torch.optim.Adam(eps=1e-5).step() + LinearLR.step()
// (xuance/torch/learners/policy_gradient/ppo_learner.py:18-22,61-67; dqn_learner.py:18-22,47-53;
//  base/marl_learner.py:64-75 with multi_agent_rl/qmix_learner.py:88-96).
// The optimiser state (step counters, learning rate) lives in device memory so a captured hipGraph advances it.
#include "common.h"
#include "split3.h"

namespace xrl {

constexpr int RED_THREADS = 256;

// grad[p] = sum_s slabs[s][p]   (fixed order s = 0..S-1 -> deterministic);  block partial of sum grad^2 (fp64).
// Each thread owns 4 consecutive parameters (one 16-byte load per slab) and keeps up to 8 slab loads in flight.
// fold: slab columns [fold_off, fold_off + fold_len) hold a second partial of columns [0, fold_len) (the critic role's
// first-layer gradient of ppo_trunk_kernel); they are added after the main columns, slab group by slab group, in the same
// fixed order.  fold_len == 0: no fold.
__global__ void __launch_bounds__(RED_THREADS) grad_reduce_kernel(const float* __restrict__ slabs, int n_split,
                                                                  int64_t slab_stride, int64_t P,
                                                                  float* __restrict__ grad,
                                                                  double* __restrict__ sumsq_part, int64_t fold_off, int fold_len) {
    __shared__ double scratch[16];
    double sq = 0.0;
    const bool vec = ((slab_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(slabs) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(grad) & 15) == 0);
    if (vec) {
        // 256 threads = 64 parameter quads x 4 slab groups: thread (pq, sg) sums slabs sg, sg+4, sg+8, ... of its quad
        // (8 independent 16-byte loads in flight), the 4 group sums are combined through LDS in group order.  The slab
        // order inside a group is fixed, so the result is deterministic (it differs from a purely sequential s = 0..S-1
        // sum only by fp32 re-association).
        __shared__ double gsum[4][64][4];
        const int64_t P4 = P / 4;
        const int64_t st4 = slab_stride / 4;
        const int pq = threadIdx.x & 63, sg = threadIdx.x >> 6;
        // ONE slab (round 6: the dense layers of AC_CNN_Atari, 3.4 M parameters, 14.8 us): three of the four waves had nothing to load and
        // the block walked its ~13 trips one global round trip after the other.  The four waves take four consecutive trips at once;
        // wave 0 then adds their squares in trip order, so every partial sum of squares is the one the loop below would form.
        __shared__ double gsq[4][64];
        const bool one_slab = n_split == 1 && fold_len == 0;
        for (int64_t base0 = (int64_t)blockIdx.x * 64; one_slab && base0 < P4; base0 += (int64_t)gridDim.x * 64 * 4) {
            const int64_t i = base0 + (int64_t)sg * gridDim.x * 64 + pq;
            double q = 0.0;
            if (i < P4) {
                const float4 t = reinterpret_cast<const float4*>(slabs)[i];
                reinterpret_cast<float4*>(grad)[i] = t;
                q = (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
            }
            gsq[sg][pq] = q;
            __syncthreads();
            if (sg == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (base0 + (int64_t)k * gridDim.x * 64 + pq < P4) sq += gsq[k][pq];
            }
            __syncthreads();
        }
        for (int64_t base = (int64_t)blockIdx.x * 64; !one_slab && base < P4; base += (int64_t)gridDim.x * 64) {
            const int64_t i = base + pq;
            // float64 accumulators: the 256 per-workgroup partials of a parameter cancel heavily (policy-gradient terms sum to a
            // small total), and Adam turns a relative gradient error straight into a relative step error
            double gx = 0.0, gy = 0.0, gz = 0.0, gw = 0.0;
            if (i < P4) {
                const float4* src = reinterpret_cast<const float4*>(slabs) + i;
                int s = sg;
                for (; s + 28 < n_split; s += 32) {
                    float4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(s + 4 * j) * st4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { gx += v[j].x; gy += v[j].y; gz += v[j].z; gw += v[j].w; }
                }
                for (; s < n_split; s += 4) { const float4 v = src[(int64_t)s * st4]; gx += v.x; gy += v.y; gz += v.z; gw += v.w; }
                if (i * 4 < fold_len) {
                    const float4* src2 = src + fold_off / 4;
                    for (s = sg; s + 28 < n_split; s += 32) {
                        float4 v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = src2[(int64_t)(s + 4 * j) * st4];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { gx += v[j].x; gy += v[j].y; gz += v[j].z; gw += v[j].w; }
                    }
                    for (; s < n_split; s += 4) { const float4 v = src2[(int64_t)s * st4]; gx += v.x; gy += v.y; gz += v.z; gw += v.w; }
                }
            }
            gsum[sg][pq][0] = gx; gsum[sg][pq][1] = gy; gsum[sg][pq][2] = gz; gsum[sg][pq][3] = gw;
            __syncthreads();
            if (sg == 0 && i < P4) {
                double t0 = gsum[0][pq][0], t1 = gsum[0][pq][1], t2 = gsum[0][pq][2], t3 = gsum[0][pq][3];
#pragma unroll
                for (int k = 1; k < 4; ++k) { t0 += gsum[k][pq][0]; t1 += gsum[k][pq][1]; t2 += gsum[k][pq][2]; t3 += gsum[k][pq][3]; }
                const float4 t = make_float4((float)t0, (float)t1, (float)t2, (float)t3);       // rounded once
                reinterpret_cast<float4*>(grad)[i] = t;
                sq += (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
            }
            __syncthreads();
        }
        for (int64_t i = P4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
            double gd = 0.0;
            for (int s = 0; s < n_split; ++s) gd += slabs[(size_t)s * slab_stride + i];
            const float g = (float)gd;
            grad[i] = g;
            sq += (double)g * (double)g;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
            double gd = 0.0;
            for (int s = 0; s < n_split; ++s) gd += slabs[(size_t)s * slab_stride + i];
            const float g = (float)gd;
            grad[i] = g;
            sq += (double)g * (double)g;
        }
    }
    const double t = block_sum(sq, scratch);
    if (threadIdx.x == 0) sumsq_part[blockIdx.x] = t;
}

// torch._single_tensor_adam, evaluated per element with the scalar prefactors in float64 like the Python side:
//   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom
template <bool VEC4>
__global__ void __launch_bounds__(RED_THREADS) adam_step_kernel(float* __restrict__ params, float* __restrict__ grad,
                                                                float* __restrict__ m, float* __restrict__ v, int64_t P,
                                                                xrl_adam_state_t* __restrict__ st,
                                                                const double* __restrict__ sumsq_part, int n_part,
                                                                double max_norm, xrl_mirrors_t mir) {
    __shared__ double scratch[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) s += sumsq_part[i];
    const double total_norm = sqrt(block_sum(s, scratch));
    float coef = 1.f;
    if (max_norm > 0.0) {
        const double c = max_norm / (total_norm + 1e-6);       // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
        coef = (float)(c < 1.0 ? c : 1.0);
    }
    const int step = st->step + 1;
    const int k = st->sched_steps < st->total_iters ? st->sched_steps : st->total_iters;
    const double lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k / (double)st->total_iters);
    const double b1 = st->beta1, b2 = st->beta2;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)st->eps;
    const float w1 = (float)(1.0 - b1), fb2 = (float)b2, w2 = (float)(1.0 - b2), wd = (float)st->weight_decay;

    // one element: clipped gradient, moments, parameter (the statements of the scalar loop; every path below goes through them)
    auto element = [&](float gi, float pi, float& mi, float& vi, float& gc) -> float {
        float g = gi * coef;
        gc = g;                                                  // p.grad holds the clipped gradient afterwards
        if (wd != 0.f) g += wd * pi;
        const float mn = mi + (g - mi) * w1;
        const float vn = vi * fb2 + w2 * g * g;
        mi = mn; vi = vn;
        const float denom = sqrtf(vn) / bc2_sqrt + eps;
        return pi - step_size * (mn / denom);
    };
    const bool sync_target = mir.target && mir.target_every > 0 && step % mir.target_every == 0;
    if (VEC4) {
        // four consecutive elements per thread through 16-byte accesses (round 6: the 1.7 M parameters of AC_CNN_Atari took 29.5 us with
        // 4-byte accesses -- 47 MB of moments, parameters and mirrors per step); mirror stores stay scattered 4-byte stores
        const int64_t P4 = P >> 2;
        for (int64_t q4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q4 < P4; q4 += (int64_t)gridDim.x * blockDim.x) {
            float4 g4 = reinterpret_cast<const float4*>(grad)[q4], p4 = reinterpret_cast<const float4*>(params)[q4];
            float4 m4 = reinterpret_cast<const float4*>(m)[q4], v4 = reinterpret_cast<const float4*>(v)[q4];
            int4 mp[XRL_MAX_MIRRORS];
#pragma unroll
            for (int q = 0; q < XRL_MAX_MIRRORS; ++q) if (q < mir.n) mp[q] = reinterpret_cast<const int4*>(mir.map[q])[q4];
            float4 gc, pn;
            pn.x = element(g4.x, p4.x, m4.x, v4.x, gc.x); pn.y = element(g4.y, p4.y, m4.y, v4.y, gc.y);
            pn.z = element(g4.z, p4.z, m4.z, v4.z, gc.z); pn.w = element(g4.w, p4.w, m4.w, v4.w, gc.w);
            reinterpret_cast<float4*>(grad)[q4] = gc;
            reinterpret_cast<float4*>(m)[q4] = m4; reinterpret_cast<float4*>(v)[q4] = v4;
            reinterpret_cast<float4*>(params)[q4] = pn;
#pragma unroll
            for (int q = 0; q < XRL_MAX_MIRRORS; ++q)
                if (q < mir.n) {
                    mirror_store(mir.dst[q], mp[q].x, pn.x, mir.split_plane);
                    mirror_store(mir.dst[q], mp[q].y, pn.y, mir.split_plane);
                    mirror_store(mir.dst[q], mp[q].z, pn.z, mir.split_plane);
                    mirror_store(mir.dst[q], mp[q].w, pn.w, mir.split_plane);
                }
            if (sync_target) {
                reinterpret_cast<float4*>(mir.target)[q4] = pn;
                if (mir.target_image) {
                    const int4 j = mir.n > 0 ? mp[0] : reinterpret_cast<const int4*>(mir.map[0])[q4];
                    if (j.x >= 0) mir.target_image[j.x] = pn.x;
                    if (j.y >= 0) mir.target_image[j.y] = pn.y;
                    if (j.z >= 0) mir.target_image[j.z] = pn.z;
                    if (j.w >= 0) mir.target_image[j.w] = pn.w;
                }
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
            float mi = m[i], vi = v[i], gc;
            const float pn = element(grad[i], params[i], mi, vi, gc);
            grad[i] = gc; m[i] = mi; v[i] = vi;
            params[i] = pn;
            // derived layouts kept in sync in the same launch (transposed middle weights, packed LDS-cache image)
#pragma unroll
            for (int q = 0; q < XRL_MAX_MIRRORS; ++q)
                if (q < mir.n) mirror_store(mir.dst[q], mir.map[q][i], pn, mir.split_plane);
            if (sync_target) {
                mir.target[i] = pn;
                if (mir.target_image) { const int j = mir.map[0][i]; if (j >= 0) mir.target_image[j] = pn; }
            }
        }
    }
    // The last block to finish advances the device-resident state (every block has consumed the old state by
    // the time it takes its ticket; the next launch observes the new state across the kernel boundary).
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ticket = atomicAdd(&st->ticket, 1);
        if (ticket == (int)gridDim.x - 1) {
            st->ticket = 0;
            st->last_grad_norm = total_norm;
            st->step = step;
            const int ns = st->sched_steps + 1;                  // scheduler.step() after optimizer.step()
            st->sched_steps = ns;
            const int k2 = ns < st->total_iters ? ns : st->total_iters;
            st->last_lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k2 / (double)st->total_iters);
        }
    }
}

// grad_reduce + adam_step in ONE launch (same arithmetic, same summation orders, bit-identical results): block b owns
// parameters [256 b, 256 b + 256) in both phases, keeps the reduced gradient on chip, publishes its sum of squares with a
// device-scope store and meets the other blocks at a counter barrier (relaxed atomics; all blocks are resident: one
// 256-thread block per 256 parameters).  Saves a kernel boundary, the argument fetch + first round trip of the second
// launch and the re-read of the gradient.  sync[1] departures (self-resetting), sync[2] time-out flag, sync[4 + b] arrival flag of block b.
// 256-thread groups per block of reduce_adam_kernel: 1 while the blocks do not outnumber the CUs (measured at the headline's 134
// blocks: 4 groups = 4x fewer barrier participants, but the 35 MB slab read then rides on 34 CUs: 22.8 vs 15.8 us); 2 once
// that still leaves a block for every CU (the MuJoCo network's 558 blocks -> 279: half as many flags to publish and poll).
constexpr int RA_GROUPS_WIDE = 2;

// block_sum over ONE 256-thread group of a larger block (same tree as block_sum on a 256-thread block)
__device__ __forceinline__ double group_sum(double v, double* scratch4, int tg) {
    v = wave_sum(v);
    const int lane = tg & 63, w = tg >> 6;
    __syncthreads();
    if (lane == 0) scratch4[w] = v;
    __syncthreads();
    double t = (lane < 4) ? scratch4[lane] : 0.0;
    return wave_sum(t);
}

// XC: the ranks of a data-parallel job meet INSIDE this launch.  Each group publishes its 256 reduced gradient values in its
// rank's exchange buffer (fine-grained device memory every peer has mapped through an IPC handle), flags them with the
// optimiser step number, waits for the same flag of every peer, reads their 256 values over xGMI and averages in rank
// order -- so every rank holds bit-identical averaged gradients and the launch count of an update equals the single-GPU
// one (no collective call, nothing for a graph to be cut at).  Buffers alternate with the step's parity: a peer can only
// publish step s + 1 after its step-s launch -- and with it every read of this rank's step-s values -- has finished.
template <bool XC, int G>
__global__ void __launch_bounds__(RED_THREADS * G) reduce_adam_kernel(const float* __restrict__ slabs, int n_split, int64_t slab_stride,
                                                                 float* __restrict__ params, float* __restrict__ grad,
                                                                 float* __restrict__ m, float* __restrict__ v, int64_t P,
                                                                 xrl_adam_state_t* __restrict__ st, double* sumsq_part, int n_part,
                                                                 double max_norm, xrl_mirrors_t mir, unsigned* sync,
                                                                 xrl_exchange_t xc) {
    // a block = G groups of 256 threads; group `vb` (virtual block) does what block vb of grad_reduce_kernel /
    // adam_step_kernel does, so every partial sum and every parameter sees the same arithmetic; fewer, larger blocks keep
    // the number of barrier participants (device-scope atomics) small.
    __shared__ double scratch[16];
    __shared__ double gscratch[G][4];
    __shared__ double gsum[G][4][64][4];
    __shared__ float gtot[G][256];
    __shared__ int s_fail, s_xfail;
    const int grp = threadIdx.x >> 8, tg = threadIdx.x & 255;
    const int vb = blockIdx.x * G + grp, n_vb = (int)((P / 4 + 63) / 64);
    const unsigned failed_before = __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an earlier launch timed out
    if (XC && threadIdx.x == 0) s_xfail = 0;
    // bookkeeping of the update phase that used to be launches of their own (xrl_mirrors_t.tick / .part): the last block
    // (the one with the fewest parameters) does it while its slab loads are in flight
    if (blockIdx.x == gridDim.x - 1) {
        if (mir.part_out && threadIdx.x < 8) {
            const double* pp = mir.part + threadIdx.x;
            double s = 0.0;
            int r = 0;
            for (; r + 8 <= mir.part_rows; r += 8) {                     // xrl_sum_partials' order: row by row
                double w8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w8[q] = pp[(size_t)(r + q) * 8];
#pragma unroll
                for (int q = 0; q < 8; ++q) s += w8[q];
            }
            for (; r < mir.part_rows; ++r) s += pp[(size_t)r * 8];
            mir.part_out[threadIdx.x] = s;
        }
        if (mir.tick && threadIdx.x == 64) *mir.tick += (unsigned)mir.tick_inc;
    }
    const int64_t P4 = P / 4, st4 = slab_stride / 4;
    const int pq = tg & 63, sg = tg >> 6;
    const int64_t qi = (int64_t)vb * 64 + pq;
    // ---- phase 1: exactly grad_reduce_kernel's vector path for this group's 64 quads
    double sq = 0.0;
    {
        double gx = 0.0, gy = 0.0, gz = 0.0, gw = 0.0;                   // (same statements as grad_reduce_kernel)
        if (qi < P4) {
            const float4* src = reinterpret_cast<const float4*>(slabs) + qi;
            int s = sg;
            if (mir.alt_split > 0) {                                     // ranges whose gradient is split over fewer rows (xrl_wide_dw1)
                const int64_t e0 = qi * 4;
                if ((e0 >= mir.alt_lo[0] && e0 < mir.alt_hi[0]) || (e0 >= mir.alt_lo[1] && e0 < mir.alt_hi[1])) n_split = mir.alt_split;
            }
            // (same summation order whatever the batching: 32 loads in flight per thread turn the four dependent round trips of a
            //  128-slab reduction into one)
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
        gsum[grp][sg][pq][0] = gx; gsum[grp][sg][pq][1] = gy; gsum[grp][sg][pq][2] = gz; gsum[grp][sg][pq][3] = gw;
        __syncthreads();
        if (sg == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi < P4) {
                double t0 = gsum[grp][0][pq][0], t1 = gsum[grp][0][pq][1], t2 = gsum[grp][0][pq][2], t3 = gsum[grp][0][pq][3];
#pragma unroll
                for (int k = 1; k < 4; ++k) { t0 += gsum[grp][k][pq][0]; t1 += gsum[grp][k][pq][1]; t2 += gsum[grp][k][pq][2]; t3 += gsum[grp][k][pq][3]; }
                t = make_float4((float)t0, (float)t1, (float)t2, (float)t3);
            }
            if (XC) {                                                    // (sg == 0: exactly one wave per group gets here)
                const unsigned xstep = (unsigned)(st->step + 1);
                const int64_t slot = (int64_t)(xstep & 1u) * xc.stride4 + qi;
                const int64_t fslot = (int64_t)(xstep & 1u) * XRL_XC_MAX_GROUPS + vb;
                if (qi < P4) reinterpret_cast<float4*>(xc.base[xc.rank] + XRL_XC_DATA_OFFSET)[slot] = t;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // the wave's stores are visible system-wide ...
                if (pq == 0 && vb < n_vb)                                 // ... before its flag is
                    __hip_atomic_store(reinterpret_cast<unsigned*>(xc.base[xc.rank]) + fslot, xstep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (vb < n_vb) {
                    int spins = 0;
                    for (;;) {
                        int ok = 1;
                        if (pq < xc.world && pq != xc.rank)
                            ok = __hip_atomic_load(reinterpret_cast<const unsigned*>(xc.base[pq]) + fslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == xstep;
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > xc.max_spins) {
                            if (pq == 0) { s_xfail = 1; __hip_atomic_store(&sync[2], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                    if (qi < P4) {
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int r = 0; r < xc.world; ++r) {              // rank order on every rank
                            float4 w = t;
                            if (r != xc.rank) w = reinterpret_cast<const float4*>(xc.base[r] + XRL_XC_DATA_OFFSET)[slot];
                            if (r == 0) a = w; else { a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w; }
                        }
                        t = make_float4(a.x * xc.inv_world, a.y * xc.inv_world, a.z * xc.inv_world, a.w * xc.inv_world);
                    }
                }
            }
            if (qi < P4) sq += (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
            *reinterpret_cast<float4*>(&gtot[grp][pq * 4]) = t;
        }
    }
    const double tsum = group_sum(sq, gscratch[grp], tg);
    // optimiser scalars and this thread's parameter / moments while the barrier is crossed
    const int step = st->step + 1;
    const int k = st->sched_steps < st->total_iters ? st->sched_steps : st->total_iters;
    const double lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k / (double)st->total_iters);
    const double b1 = st->beta1, b2 = st->beta2;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)st->eps;
    const float w1 = (float)(1.0 - b1), fb2 = (float)b2, w2 = (float)(1.0 - b2), wd = (float)st->weight_decay;
    const int64_t i = (int64_t)vb * RED_THREADS + tg;
    float p0 = 0.f, m0 = 0.f, v0 = 0.f;
    int mj[XRL_MAX_MIRRORS];                                          // (the mirror slots of parameter i: fetched here too, not
#pragma unroll                                                        //  as a dependent load behind the barrier)
    for (int q = 0; q < XRL_MAX_MIRRORS; ++q) mj[q] = -1;
    if (i < P) {
        p0 = params[i]; m0 = m[i]; v0 = v[i];
#pragma unroll
        for (int q = 0; q < XRL_MAX_MIRRORS; ++q) if (q < mir.n) mj[q] = mir.map[q][i];
    }
    // ---- barrier without read-modify-write atomics: every block publishes its partial sum, then (after the store has been
    //      acknowledged) its flag = the optimiser step this launch performs -- a value no earlier launch has written -- and
    //      polls all flags with one coalesced device-scope load per round.  sync[4 + b] is block b's flag.
    //      Without clipping (max_norm <= 0: configs/qmix/sc2/3m.yaml:45, configs/dqn/atari.yaml:39) nothing below depends on
    //      the other blocks: no barrier at all; the last block out still reports the norm from the published partial sums.
    const bool need_norm = max_norm > 0.0;
    // (a block whose peer wait expired publishes NaN: with clipping on, every block of this rank then sees a NaN norm and steps nothing)
    if (tg == 0 && vb < n_vb) __hip_atomic_store(&sumsq_part[vb], (XC && s_xfail) ? __builtin_nan("") : tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        s_fail = 0;
        if (need_norm) __hip_atomic_store(&sync[4 + blockIdx.x], (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (need_norm) {
        // (with the ranks meeting in this launch a block reaches this barrier only after ITS wait for the peers' rows: the blocks that
        //  were served first must outwait the ones still waiting for a late rank, so the bound grows with the exchange's own -- a spin
        //  here is several times shorter than one there; seen with four rank processes time-sharing one GPU: the barrier expired while
        //  other blocks of the same launch were still inside their (much longer) peer wait)
        const int barrier_spins = !XC ? 2000000 : (xc.max_spins > 500000000 ? 2147000000 : 2000000 + 4 * xc.max_spins);
        int spins = 0;
        for (;;) {
            int ok = 1;
            for (int j = threadIdx.x; j < (int)gridDim.x; j += blockDim.x)
                ok &= __hip_atomic_load(&sync[4 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)step;
            if (__syncthreads_and(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > barrier_spins) { if (threadIdx.x == 0) { s_fail = 1; __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } break; }
        }
    }
    __syncthreads();
    // ---- phase 2: exactly adam_step_kernel for parameter i (every group forms the norm like a 256-thread block would)
    double ssum = 0.0;
    if (need_norm)
        for (int j = tg; j < n_part; j += RED_THREADS)
            ssum += j < n_vb ? __hip_atomic_load(&sumsq_part[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    double total_norm = (s_fail || (XC && s_xfail)) ? __builtin_nan("") : (need_norm ? sqrt(group_sum(ssum, gscratch[grp], tg)) : 0.0);
    float coef = 1.f;
    if (max_norm > 0.0) {
        const double c = max_norm / (total_norm + 1e-6);
        coef = (float)(c < 1.0 ? c : 1.0);
    }
    // a failed wait (barrier or peer exchange) must not step anything, whatever max_norm is: with max_norm <= 0 the NaN norm
    // above is never multiplied in, and Adam on a partial / stale peer average would leave the replicas diverged silently.
    // The block that failed skips its parameters; sync[2] stays set and the last block out reports a NaN norm (the learners
    // read both after the phase and raise).
    // (with clipping off there is no barrier in front of this point, so a failed wait in ANOTHER block, or in an earlier
    //  launch of the same captured phase, is seen through the status word itself: one relaxed load per thread, issued at kernel start)
    const bool poisoned = s_fail == 1 || (XC && s_xfail) || total_norm != total_norm || failed_before != 0u;
    if (i < P && !poisoned) {
        float g = gtot[grp][tg] * coef;
        grad[i] = g;
        if (wd != 0.f) g += wd * p0;
        const float mi = m0 + (g - m0) * w1;
        const float vi = v0 * fb2 + w2 * g * g;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pn = p0 - step_size * (mi / denom);
        params[i] = pn;
#pragma unroll
        for (int q = 0; q < XRL_MAX_MIRRORS; ++q)
            if (q < mir.n) mirror_store(mir.dst[q], mj[q], pn, mir.split_plane);
        if (mir.target && mir.target_every > 0 && step % mir.target_every == 0) {
            mir.target[i] = pn;
            if (mir.target_image && mj[0] >= 0) mir.target_image[mj[0]] = pn;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (need_norm) {
            // every block read the optimiser state BEFORE it published its flag, so whoever is past the barrier may advance
            // it: block 0 does, and nobody takes a ticket (one read-modify-write on ONE address per block serialises: with
            // the 558 blocks of the MuJoCo network that chain was several microseconds long)
            s_fail = blockIdx.x == 0 ? 2 : 0;
        } else {
            const unsigned left = __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_fail = (left == gridDim.x - 1) ? 2 : 0;            // (re-used as "this is the last block out")
        }
    }
    __syncthreads();
    if (s_fail == 2) {                                          // last block out: reset the counter, advance the state
        if (!need_norm) {                                       // every block published its partial before its ticket
            double t = 0.0;
            for (int j = threadIdx.x; j < n_vb; j += blockDim.x)
                t += __hip_atomic_load(&sumsq_part[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total_norm = sqrt(block_sum(t, scratch));
            if (__hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) total_norm = __builtin_nan("");
        }
        if (threadIdx.x == 0) {
            __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st->last_grad_norm = total_norm;
            st->step = step;
            const int ns = st->sched_steps + 1;
            st->sched_steps = ns;
            const int k2 = ns < st->total_iters ? ns : st->total_iters;
            st->last_lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k2 / (double)st->total_iters);
        }
    }
}

}  // namespace xrl

using namespace xrl;

// How many blocks of the launch xrl_reduce_adam would make for P parameters can be RESIDENT at once (its barrier spins, so all of
// them must be): the runtime's occupancy figure for the kernel instance that would run, times the compute units.  An
// otherwise idle device is assumed -- another process on the same GPU (test boxes) can still take CUs away, which the
// barrier's time-out reports (sync[2]); config.use_fused_optimizer: False selects the two-launch sequence.
static void reduce_adam_geometry(int64_t P, bool exchange, int* n_blocks, int* capacity) {
    const int n_vb = (int)((P / 4 + 63) / 64);
    const int cus = device_cu_count();
    const int G = n_vb >= RA_GROUPS_WIDE * cus ? RA_GROUPS_WIDE : 1;
    *n_blocks = (n_vb + G - 1) / G;
    static int per_cu[2][2] = {{0, 0}, {0, 0}};
    int& pc = per_cu[exchange ? 1 : 0][G > 1 ? 1 : 0];
    if (pc == 0) {
        int nb = 0;
        const void* fn = exchange ? (G > 1 ? (const void*)reduce_adam_kernel<true, RA_GROUPS_WIDE> : (const void*)reduce_adam_kernel<true, 1>)
                                  : (G > 1 ? (const void*)reduce_adam_kernel<false, RA_GROUPS_WIDE> : (const void*)reduce_adam_kernel<false, 1>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, RED_THREADS * G, 0) != hipSuccess || nb < 1) nb = 1;
        pc = nb;
    }
    *capacity = pc * cus;
}

extern "C" int xrl_reduce_adam_fits(int64_t P, int with_exchange) {
    if (P <= 0 || (P & 3)) return 0;
    int nb = 0, cap = 0;
    reduce_adam_geometry(P, with_exchange != 0, &nb, &cap);
    return (nb <= cap && (P / 4 + 63) / 64 <= 1024) ? 1 : 0;
}

extern "C" int xrl_reduce_adam_exchange(const float* slabs, int n_split, int64_t slab_stride, float* params, float* grad,
                                        float* m, float* v, int64_t P, xrl_adam_state_t* state, double* sumsq_part, int n_part,
                                        double max_norm, const xrl_mirrors_t* mirrors, uint32_t* sync,
                                        const xrl_exchange_t* exchange, xrl_stream_t stream) {
    XRL_CHECK_ARG(slabs && params && grad && m && v && state && sumsq_part && sync && n_split >= 1 && P > 0);
    XRL_CHECK_ARG((P & 3) == 0 && (slab_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(slabs) & 15) == 0));
    const int n_vb = (int)((P / 4 + 63) / 64);
    XRL_CHECK_ARG(n_vb <= n_part && n_part <= 1024);
    const int G = n_vb >= RA_GROUPS_WIDE * device_cu_count() ? RA_GROUPS_WIDE : 1;   // (every CU keeps a block)
    const int nb = (n_vb + G - 1) / G;
    {
        int nb_q = 0, cap = 0;                                  // every block resident (the barrier spins): asked of the runtime
        reduce_adam_geometry(P, exchange && exchange->world > 1, &nb_q, &cap);
        if (nb > cap) {
            set_error("xrl_reduce_adam: the launch's blocks cannot all be resident on this device (its barrier spins); use "
                      "xrl_grad_reduce + xrl_adam_step (xrl_reduce_adam_fits tells in advance)");
            return XRL_EINVAL;
        }
    }
    xrl_mirrors_t mir{};
    if (mirrors) mir = *mirrors;
    XRL_CHECK_ARG(mir.n >= 0 && mir.n <= XRL_MAX_MIRRORS && mir.split_plane >= 0);
    for (int q = 0; q < mir.n; ++q) XRL_CHECK_ARG(mir.map[q] && mir.dst[q]);
    XRL_CHECK_ARG(mir.target_image == nullptr || (mir.n >= 1 && mir.target));
    XRL_CHECK_ARG(mir.fold_len >= 0 && (mir.fold_len & 3) == 0 && (mir.fold_off & 3) == 0 &&
                  (mir.fold_len == 0 || (mir.fold_off >= P && mir.fold_off + mir.fold_len <= slab_stride && mir.fold_len <= P)));
    XRL_CHECK_ARG((mir.part == nullptr) == (mir.part_out == nullptr) && (mir.part == nullptr || mir.part_rows >= 1));
    XRL_CHECK_ARG(mir.alt_split >= 0 && mir.alt_split <= n_split);
    if (mir.alt_split > 0)
        for (int i = 0; i < 2; ++i) XRL_CHECK_ARG(mir.alt_lo[i] >= 0 && mir.alt_lo[i] <= mir.alt_hi[i] && mir.alt_hi[i] <= P && !(mir.alt_lo[i] & 3) && !(mir.alt_hi[i] & 3));
    if (exchange && exchange->world > 1) {
        const xrl_exchange_t& xc = *exchange;
        XRL_CHECK_ARG(xc.world <= XRL_XC_MAX_RANKS && xc.rank >= 0 && xc.rank < xc.world && n_vb <= XRL_XC_MAX_GROUPS);
        XRL_CHECK_ARG(xc.stride4 >= P / 4 && xc.max_spins > 0);
        for (int r = 0; r < xc.world; ++r) XRL_CHECK_ARG(xc.base[r] != nullptr);
        if (G == 1)
            hipLaunchKernelGGL((reduce_adam_kernel<true, 1>), dim3(nb), dim3(RED_THREADS), 0, as_stream(stream), slabs, n_split, slab_stride,
                               params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir, sync, xc);
        else
            hipLaunchKernelGGL((reduce_adam_kernel<true, RA_GROUPS_WIDE>), dim3(nb), dim3(RED_THREADS * RA_GROUPS_WIDE), 0, as_stream(stream), slabs, n_split, slab_stride,
                               params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir, sync, xc);
    } else if (G == 1) {
        hipLaunchKernelGGL((reduce_adam_kernel<false, 1>), dim3(nb), dim3(RED_THREADS), 0, as_stream(stream), slabs, n_split, slab_stride,
                           params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir, sync, xrl_exchange_t{});
    } else {
        hipLaunchKernelGGL((reduce_adam_kernel<false, RA_GROUPS_WIDE>), dim3(nb), dim3(RED_THREADS * RA_GROUPS_WIDE), 0, as_stream(stream), slabs, n_split, slab_stride,
                           params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir, sync, xrl_exchange_t{});
    }
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_reduce_adam(const float* slabs, int n_split, int64_t slab_stride, float* params, float* grad, float* m,
                               float* v, int64_t P, xrl_adam_state_t* state, double* sumsq_part, int n_part, double max_norm,
                               const xrl_mirrors_t* mirrors, uint32_t* sync, xrl_stream_t stream) {
    return xrl_reduce_adam_exchange(slabs, n_split, slab_stride, params, grad, m, v, P, state, sumsq_part, n_part, max_norm,
                                    mirrors, sync, nullptr, stream);
}

// ---- exchange buffers: fine-grained device memory (coherent for system-scope accesses of peers) shared through IPC handles
extern "C" int xrl_ipc_alloc(size_t bytes, void** ptr_out, unsigned char* handle_out) {
    XRL_CHECK_ARG(bytes > 0 && ptr_out && handle_out);
    void* p = nullptr;
    XRL_CHECK_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    XRL_CHECK_HIP(hipMemset(p, 0, bytes));
    XRL_CHECK_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    XRL_CHECK_HIP(hipIpcGetMemHandle(&h, p));
    static_assert(sizeof(h) == XRL_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    memcpy(handle_out, &h, sizeof(h));
    *ptr_out = p;
    return XRL_OK;
}

extern "C" int xrl_ipc_open(const unsigned char* handle, void** ptr_out) {
    XRL_CHECK_ARG(handle && ptr_out);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    XRL_CHECK_HIP(hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess));
    return XRL_OK;
}

extern "C" int xrl_ipc_close(void* peer_ptr) {
    XRL_CHECK_ARG(peer_ptr != nullptr);
    XRL_CHECK_HIP(hipIpcCloseMemHandle(peer_ptr));
    return XRL_OK;
}

extern "C" int xrl_ipc_free(void* ptr) {
    XRL_CHECK_ARG(ptr != nullptr);
    XRL_CHECK_HIP(hipFree(ptr));
    return XRL_OK;
}

extern "C" int xrl_ipc_clear(void* ptr, size_t bytes, xrl_stream_t stream) {
    XRL_CHECK_ARG(ptr != nullptr && bytes > 0);
    XRL_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, as_stream(stream)));     // (never inside a captured graph: see rollout_persist.hip)
    return XRL_OK;
}

extern "C" int xrl_grad_reduce(const float* slabs, int n_split, int64_t slab_stride, int64_t P, float* grad,
                               double* sumsq_part, int n_part, xrl_stream_t stream) {
    XRL_CHECK_ARG(slabs && grad && sumsq_part && n_split >= 1 && P > 0 && n_part >= 1 && n_part <= 1024);
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(n_part), dim3(RED_THREADS), 0, as_stream(stream), slabs, n_split,
                       slab_stride, P, grad, sumsq_part, (int64_t)0, 0);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_grad_reduce_fold(const float* slabs, int n_split, int64_t slab_stride, int64_t P, float* grad,
                                    double* sumsq_part, int n_part, int64_t fold_off, int fold_len, xrl_stream_t stream) {
    XRL_CHECK_ARG(slabs && grad && sumsq_part && n_split >= 1 && P > 0 && n_part >= 1 && n_part <= 1024);
    XRL_CHECK_ARG((slab_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0);
    XRL_CHECK_ARG(fold_len >= 0 && (fold_len & 3) == 0 && (fold_off & 3) == 0 &&
                  (fold_len == 0 || (fold_off >= P && fold_off + fold_len <= slab_stride && fold_len <= P)));
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(n_part), dim3(RED_THREADS), 0, as_stream(stream), slabs, n_split,
                       slab_stride, P, grad, sumsq_part, fold_off, fold_len);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

// 16-byte accesses when every array allows them (P a multiple of 4, all bases 16-byte aligned), else element by element
static void launch_adam_step(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state, const double* sumsq_part,
                             int n_part, double max_norm, const xrl_mirrors_t& mir, hipStream_t stream) {
    uintptr_t bits = reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
        reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(mir.target);
    for (int q = 0; q < mir.n; ++q) bits |= reinterpret_cast<uintptr_t>(mir.map[q]);
    if (mir.target_image && mir.n == 0) bits |= 1;                      // (the target image is addressed through map[0])
    const bool vec = (P & 3) == 0 && (bits & 15) == 0;
    const int64_t units = vec ? P / 4 : P;
    int nb = (int)((units + RED_THREADS - 1) / RED_THREADS);
    if (nb > 1024) nb = 1024;
    if (vec) hipLaunchKernelGGL(adam_step_kernel<true>, dim3(nb), dim3(RED_THREADS), 0, stream, params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir);
    else hipLaunchKernelGGL(adam_step_kernel<false>, dim3(nb), dim3(RED_THREADS), 0, stream, params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir);
}

extern "C" int xrl_adam_step(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state,
                             const double* sumsq_part, int n_part, double max_norm, xrl_stream_t stream) {
    XRL_CHECK_ARG(params && grad && m && v && state && sumsq_part && P > 0 && n_part >= 1 && n_part <= 1024);
    launch_adam_step(params, grad, m, v, P, state, sumsq_part, n_part, max_norm, xrl_mirrors_t{}, as_stream(stream));
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_adam_step_mirrored(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state,
                                      const double* sumsq_part, int n_part, double max_norm, const int32_t* map_a,
                                      float* dst_a, const int32_t* map_b, float* dst_b, xrl_stream_t stream) {
    XRL_CHECK_ARG(params && grad && m && v && state && sumsq_part && P > 0 && n_part >= 1 && n_part <= 1024);
    XRL_CHECK_ARG((!map_a || dst_a) && (!map_b || dst_b));
    xrl_mirrors_t mir{};
    if (map_a) { mir.map[mir.n] = map_a; mir.dst[mir.n] = dst_a; ++mir.n; }
    if (map_b) { mir.map[mir.n] = map_b; mir.dst[mir.n] = dst_b; ++mir.n; }
    return xrl_adam_step_mirrors(params, grad, m, v, P, state, sumsq_part, n_part, max_norm, &mir, stream);
}

extern "C" int xrl_adam_step_mirrors(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state,
                                     const double* sumsq_part, int n_part, double max_norm, const xrl_mirrors_t* mirrors,
                                     xrl_stream_t stream) {
    XRL_CHECK_ARG(params && grad && m && v && state && sumsq_part && P > 0 && n_part >= 1 && n_part <= 1024);
    xrl_mirrors_t mir{};
    if (mirrors) mir = *mirrors;
    XRL_CHECK_ARG(mir.n >= 0 && mir.n <= XRL_MAX_MIRRORS && mir.split_plane >= 0);
    for (int q = 0; q < mir.n; ++q) XRL_CHECK_ARG(mir.map[q] && mir.dst[q]);
    launch_adam_step(params, grad, m, v, P, state, sumsq_part, n_part, max_norm, mir, as_stream(stream));
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
