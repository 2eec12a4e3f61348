// Fused rollout step: ONE launch per vector step of the on-policy agent loop (ppo_agent.py:113-177) for a
// device-resident CartPole.  See include/xrl_hip.h (xrl_rollout_step_t) for the dataflow contract.
//
// Why: at the reference's sizes (256..1024 envs, 34k-parameter net) a vector step is ~17 MFLOP and ~20 KB of
// traffic -- far below one launch's worth of roofline work -- so the seven-launch unfused sequence spends its time
// on kernel boundaries (measured ~4.5 us each, profiles/r01_a_*).  Here a workgroup keeps a 32-row tile of every
// activation level in LDS, streams the (L2-resident) weights straight into MFMA B-fragments and finishes sampling,
// physics and bookkeeping for its 32 envs before exiting.
//
// MFMA mapping (64-wide wavefronts): v_mfma_f32_32x32x2_f32, the tile's 32 rows are the M dimension, each wave owns
// 32-column output tiles.  A-fragments are ds_read_b128 from the LDS activations (row stride = roundup8(width)+4
// floats: conflict-free), B-fragments are 16-byte global loads of 4 consecutive k of one weight row per lane; as in
// gemm.hip lane-half h consumes k = 8q+4h+s so both operands use the same k permutation.  Layers with fewer than 4
// column tiles (the heads) split K over the 4 waves and reduce through LDS in a fixed order.
#include "common.h"
#include "rng.h"
#include "cartpole.h"

namespace xrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FT = 32;             // rows per tile
constexpr int FUSED_THREADS = 256; // 4 waves

__host__ __device__ inline int level_ld(int width) { return ((width + 7) / 8) * 8 + 4; }

// out[32][N] (+out_off) = act(in[32][K] (+in_off) . W[N][K]^T + b)      in/out: LDS tiles, W/b: global
__device__ __forceinline__ void fused_layer(const float* __restrict__ W, const float* __restrict__ bias, int K, int N,
                                            int act, const float* in, int ld_in, float* out, int ld_out, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = (N + 31) / 32;
    const int kq = (K + 7) / 8;    // chunks of 8 k
    const bool vecW = ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    const bool split_k = n_tiles < 4;
    for (int tile = split_k ? 0 : wave; tile < n_tiles; tile += split_k ? 1 : 4) {
        const int n0 = tile * 32;
        const int wr = n0 + li;                                   // weight row of this lane's B fragment
        const float* wrow = W + (size_t)wr * K;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int q0 = split_k ? wave : 0, qs = split_k ? 4 : 1;
#pragma unroll 4
        for (int q = q0; q < kq; q += qs) {
            const int kk = q * 8 + 4 * lh;
            const float4 a = *reinterpret_cast<const float4*>(&in[li * ld_in + kk]);
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wr < N) {
                if (vecW && kk + 3 < K) b = *reinterpret_cast<const float4*>(wrow + kk);
                else {
                    if (kk + 0 < K) b.x = wrow[kk + 0];
                    if (kk + 1 < K) b.y = wrow[kk + 1];
                    if (kk + 2 < K) b.z = wrow[kk + 2];
                    if (kk + 3 < K) b.w = wrow[kk + 3];
                }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (!split_k) {
            const int col = n0 + li;
            if (col < N) {
                const float bv = bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    out[row * ld_out + col] = act_apply(acc[r] + bv, act);
                }
            }
        } else {
            // partial tiles of the 4 waves -> red[wave][row][33], summed in wave order (deterministic)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                red[(wave * 32 + row) * 33 + li] = acc[r];
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 32 * 32; i += FUSED_THREADS) {
                const int row = i >> 5, c = i & 31, col = n0 + c;
                if (col < N) {
                    float v = red[(0 * 32 + row) * 33 + c];
                    v += red[(1 * 32 + row) * 33 + c];
                    v += red[(2 * 32 + row) * 33 + c];
                    v += red[(3 * 32 + row) * 33 + c];
                    out[row * ld_out + col] = act_apply(v + bias[col], act);
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(FUSED_THREADS) rollout_step_cartpole_kernel(xrl_rollout_step_t p) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double part[FUSED_THREADS];
    __shared__ double bmean[64], bvar[64];
    __shared__ float s_mean[64], s_std[64];
    __shared__ float s_ret[2];
    __shared__ unsigned long long s_mask[64];

    const int tid = threadIdx.x, n = p.n, D = p.D, A = p.A;
    const int n_act_tiles = (n + FT - 1) / FT;
    const bool boot = p.boot_only || ((int)blockIdx.x >= n_act_tiles);
    const int tile = p.boot_only ? blockIdx.x : (boot ? blockIdx.x - n_act_tiles : blockIdx.x);
    const int e0 = tile * FT;

    // LDS carve: level buffers, then the split-K reduction scratch (offsets kept in LDS: they are indexed at run time)
    __shared__ int lvl_off[XRL_FUSED_MAX_LEVELS], lvl_ld[XRL_FUSED_MAX_LEVELS];
    int off = 0;
    for (int l = 0; l < p.n_levels; ++l) {
        const int ld = level_ld(p.level_width[l]);
        if (tid == 0) { lvl_ld[l] = ld; lvl_off[l] = off; }
        off += FT * ld;
    }
    float* red = lds + off;
    for (int i = tid; i < off; i += FUSED_THREADS) lds[i] = 0.f;      // padding columns must read as zero

    if (!boot) {
        // ---- deferred ret_rms.update() of the episodes that ended at the previous step, in env order (ppo_agent.py:146-149)
        if (tid < 64) {
            float mean = p.ret_stats_in[0], var = p.ret_stats_in[1];
            double count = *p.ret_count_in;
            for (int base = 0; base < n; base += 64) {
                const int e = base + tid;
                const unsigned long long m = __ballot(e < n && p.ended_in[e] != 0);
                unsigned long long mm = m;
                while (mm) {                                            // wave-uniform loop over finished envs
                    const int bpos = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const float bm = p.ret_final_in[base + bpos];
                    const double tot = count + 1.0;
                    const float delta = bm - mean;
                    const float new_mean = mean + delta * 1.0f / (float)tot;
                    const float m_a = var * (float)count;
                    const float M2 = m_a + 0.f + (delta * delta) * (float)count * 1.0f / (float)tot;
                    mean = new_mean; var = M2 / (float)tot; count = tot;
                }
            }
            if (tid == 0) {
                s_ret[0] = mean; s_ret[1] = var;
                if (blockIdx.x == 0) { p.ret_stats_out[0] = mean; p.ret_stats_out[1] = var; *p.ret_count_out = count; }
            }
        }
        // ---- obs_rms.update(obs) over ALL envs (recomputed per workgroup), then normalise this tile's rows
        if (p.use_obsnorm) {
            const int R = FUSED_THREADS / D;
            const int d = tid % D, r0 = tid / D;
            const bool live = r0 < R;
            double s = 0.0;
            if (live) for (int r = r0; r < n; r += R) s += (double)p.obs_raw_in[(size_t)r * D + d];
            part[tid] = live ? s : 0.0;
            __syncthreads();
            if (tid < D) {
                double t = 0.0;
                for (int r = 0; r < R; ++r) t += part[r * D + tid];
                bmean[tid] = (double)(float)(t / n);
            }
            __syncthreads();
            double q = 0.0;
            if (live) {
                const double m = bmean[d];
                for (int r = r0; r < n; r += R) { const double df = (double)p.obs_raw_in[(size_t)r * D + d] - m; q += df * df; }
            }
            part[tid] = live ? q : 0.0;
            __syncthreads();
            if (tid < D) {
                double t = 0.0;
                for (int r = 0; r < R; ++r) t += part[r * D + tid];
                const float bstd = (float)sqrt(t / n);
                const float bv = bstd * bstd, bm = (float)bmean[tid];
                const double cnt = *p.obs_count_in, tot = cnt + (double)n;
                const float mean = p.obs_stats_in[tid], var = p.obs_stats_in[D + tid];
                const float delta = bm - mean;
                const float new_mean = mean + delta * (float)n / (float)tot;
                const float m_a = var * (float)cnt, m_b = bv * (float)n;
                const float M2 = m_a + m_b + (delta * delta) * (float)cnt * (float)n / (float)tot;
                const float new_var = M2 / (float)tot;
                s_mean[tid] = new_mean; s_std[tid] = sqrtf(new_var);
                if (blockIdx.x == 0) {
                    p.obs_stats_out[tid] = new_mean; p.obs_stats_out[D + tid] = new_var;
                    if (tid == 0) *p.obs_count_out = tot;
                }
            }
        } else if (tid < D) { s_mean[tid] = 0.f; s_std[tid] = 1.f; }
        __syncthreads();
        for (int i = tid; i < FT * D; i += FUSED_THREADS) {
            const int r = i / D, d = i - r * D, e = e0 + r;
            float v = 0.f;
            if (e < n) {
                v = p.obs_raw_in[(size_t)e * D + d];
                if (p.use_obsnorm) { v = (v - s_mean[d]) / (s_std[d] + 1e-8f); v = fminf(fmaxf(v, -p.obs_range), p.obs_range); }
                p.obs_slot[(size_t)e * D + d] = v;                      // memory.observations[t] (ppo_agent.py:128)
            }
            lds[lvl_off[0] + r * lvl_ld[0] + d] = v;
        }
    } else {
        __syncthreads();
        for (int i = tid; i < FT * D; i += FUSED_THREADS) {
            const int r = i / D, d = i - r * D, e = e0 + r;
            lds[lvl_off[0] + r * lvl_ld[0] + d] = (e < n) ? p.xnext_in[(size_t)e * D + d] : 0.f;
        }
    }
    __syncthreads();

    // ---- the whole actor-critic MLP on the LDS-resident tile
    for (int li = 0; li < p.n_layers; ++li) {
        const xrl_fused_layer_t& L = p.layers[li];
        fused_layer(p.params + L.w_off, p.params + L.b_off, L.K, L.N, L.act,
                    lds + lvl_off[L.in_level] + L.in_off, lvl_ld[L.in_level],
                    lds + lvl_off[L.out_level] + L.out_off, lvl_ld[L.out_level], red);
    }
    const float* heads = lds + lvl_off[p.n_levels - 1];
    const int ldh = lvl_ld[p.n_levels - 1];

    if (tid >= FT) return;
    const int e = e0 + tid;
    if (e >= n) return;
    const float* h = heads + tid * ldh;
    if (boot) {                                                         // V(next_obs_{t-1}) -> bootv[t-1]
        if (p.bootv_prev) p.bootv_prev[e] = h[A];
        return;
    }
    // ---- get_actions (core/on_policy.py:128-169): sample, log-prob, value; store (ppo_agent.py:128)
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    int a = 0;
    float logp;
    {
        uint32_t r[4];
        philox4x32(p.seed, (uint32_t)e, step, STREAM_ACTION, r);
        const float u = u01(r[0]);
        float mx = h[0];
        for (int j = 1; j < A; ++j) mx = fmaxf(mx, h[j]);
        float se = 0.f;
        for (int j = 0; j < A; ++j) se += expf(h[j] - mx);
        const float lse = mx + logf(se);
        a = A - 1;
        float c = 0.f;
        for (int j = 0; j < A; ++j) { c += expf(h[j] - lse); if (c > u) { a = j; break; } }
        logp = h[a] - lse;
    }
    p.act_slot[e] = (float)a;
    p.val_slot[e] = h[A];
    p.logp_slot[e] = logp;
    // ---- envs.step(acts): physics + DummyVecEnv auto-reset
    double* s = p.cp_state + (size_t)e * 4;
    double x, xd, th, thd;
    bool term;
    cartpole_advance(s, a, x, xd, th, thd, term);
    const int steps = p.cp_steps[e] + 1;
    const bool trunc = steps >= p.max_steps;
    const float nobs[4] = {(float)x, (float)xd, (float)th, (float)thd};
    const float score = p.cp_score[e] + 1.0f;
    float robs[4] = {nobs[0], nobs[1], nobs[2], nobs[3]};
    if (term || trunc) {
        const int ep = p.cp_episodes[e] + 1;
        p.cp_episodes[e] = ep;
        cartpole_reset(s, p.env_seed, e, (uint32_t)ep);
        p.cp_steps[e] = 0; p.cp_score[e] = 0.f;
        robs[0] = (float)s[0]; robs[1] = (float)s[1]; robs[2] = (float)s[2]; robs[3] = (float)s[3];
        atomicAdd(&p.cp_stats[0], 1.0); atomicAdd(&p.cp_stats[1], (double)score); atomicAdd(&p.cp_stats[2], (double)steps);
    } else {
        s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
        p.cp_steps[e] = steps; p.cp_score[e] = score;
    }
    // ---- bookkeeping (ppo_agent.py:128,144-157)
    const float reward = 1.0f;
    float rstd = sqrtf(s_ret[1]);
    rstd = fminf(fmaxf(rstd, 0.1f), 100.f);
    float rn = reward;
    if (p.use_rewnorm) rn = fminf(fmaxf(reward / rstd, -p.rew_range), p.rew_range);
    p.rew_slot[e] = rn;
    p.term_slot[e] = term ? 1.f : 0.f;
    p.seg_slot[e] = (term || trunc || p.last_step) ? (uint8_t)(1 | (term ? 2 : 0)) : (uint8_t)0;
    const float tr = p.gamma * p.ret_track[e] + reward;
    if (term || trunc) { p.ret_final_out[e] = tr; p.ended_out[e] = 1; p.ret_track[e] = 0.f; }
    else { p.ended_out[e] = 0; p.ret_track[e] = tr; }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        p.obs_raw_out[(size_t)e * 4 + d] = robs[d];                    // buf_obs for the next step (reset_obs on episode end)
        float v = nobs[d];
        if (p.use_obsnorm) { v = (v - s_mean[d]) / (s_std[d] + 1e-8f); v = fminf(fmaxf(v, -p.obs_range), p.obs_range); }
        p.xnext_out[(size_t)e * 4 + d] = v;                            // get_terminated_values input (on_policy.py:109)
    }
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_init(void) {
    // kernels that carve more than 64 KB of dynamic LDS need the attribute; set it outside any graph capture
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_step_cartpole_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return XRL_OK;
}

extern "C" int xrl_rollout_step_cartpole(const xrl_rollout_step_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_rollout_step_t& p = *pp;
    XRL_CHECK_ARG(p.params && p.n > 0 && p.D == 4 && p.A >= 2 && !p.gaussian);
    XRL_CHECK_ARG(p.n_layers >= 1 && p.n_layers <= XRL_FUSED_MAX_LAYERS && p.n_levels >= 2 && p.n_levels <= XRL_FUSED_MAX_LEVELS);
    XRL_CHECK_ARG(p.level_width[0] == p.D && p.level_width[p.n_levels - 1] >= p.A + 1);
    XRL_CHECK_ARG(p.xnext_in != nullptr);
    if (!p.boot_only) {
        XRL_CHECK_ARG(p.obs_raw_in && p.obs_raw_out && p.xnext_out && p.obs_stats_in && p.obs_stats_out && p.obs_count_in &&
                      p.obs_count_out && p.ret_stats_in && p.ret_stats_out && p.ret_count_in && p.ret_count_out &&
                      p.ended_in && p.ended_out && p.ret_final_in && p.ret_final_out && p.ret_track);
        XRL_CHECK_ARG(p.obs_slot && p.act_slot && p.val_slot && p.logp_slot && p.rew_slot && p.term_slot && p.seg_slot);
        XRL_CHECK_ARG(p.cp_state && p.cp_steps && p.cp_episodes && p.cp_score && p.cp_stats);
    } else {
        XRL_CHECK_ARG(p.bootv_prev != nullptr);
    }
    size_t floats = 0;
    for (int l = 0; l < p.n_levels; ++l) floats += (size_t)FT * level_ld(p.level_width[l]);
    floats += 4 * 32 * 33;                                            // split-K reduction scratch
    const size_t lds_bytes = floats * sizeof(float);
    XRL_CHECK_ARG(lds_bytes <= 150 * 1024);
    const int n_tiles = (p.n + FT - 1) / FT;
    const int grid = p.boot_only ? n_tiles : (p.bootv_prev ? 2 * n_tiles : n_tiles);
    hipLaunchKernelGGL(rollout_step_cartpole_kernel, dim3(grid), dim3(FUSED_THREADS), lds_bytes, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
