// Persistent rollout: ALL T vector steps of an on-policy rollout (plus the final bootstrap pass) in ONE launch.
// Same arithmetic and bit-identical results as T launches of rollout_step_fast_kernel (rollout_fast.hip) + one
// bootstrap-only launch -- tested -- for the shape class 4-128-{128-2,128-1} and n_envs <= 320.
//
// Why (measured): a per-step launch pays, before its first useful instruction, a kernel boundary
// (1.6 us), the argument fetch (2.7 k cycles) and a memory round trip through an L2 that is invalidated at every launch
// (~4 k cycles), then re-streams its 64 KB of weights: ~5 of the 11 us of a step.  Here
//   * the workgroups stay resident: weights live in registers for the whole rollout, the per-env simulator state in
//     registers / LDS, the running statistics in registers (every workgroup computes the same update redundantly, as
//     before), the instruction cache stays warm;
//   * the only cross-workgroup traffic per step is what couples the envs: raw observations (statistics), normalised
//     next observations (bootstrap tiles), episode-end flags and returns (return statistics) -- ping-ponged like before;
//   * steps are separated by a flag barrier in L2: one relaxed store per workgroup + one coalesced polling load per round
//     (a counter with atomic adds costs ~2.2 k cycles for 24 workgroups, tools/microbench_xcd_barrier.py).  Workgroups are dealt round-robin to the 8
//     XCDs, so the grid is 8x oversubscribed and only blockIdx % 8 == 0 stays: the survivors share ONE L2, which makes
//     plain stores (write-through to L2, completed by s_waitcnt vmcnt(0)) + device-scope loads a coherent exchange
//     without any L2 write-back / invalidate.  That placement is CHECKED in every launch: each survivor publishes its XCC id
//     behind a start-up barrier; if the ids differ (partitioned device, a second process on the GPU, another dispatch order)
//     the launch switches its exchange stores to device-scope atomic stores -- correct on any placement, slower per step
//     (status[3] counts such launches, status[2] is the mask of every XCC ever seen).  A barrier time-out (status[0] = 1)
//     makes every workgroup leave; the host reads status at its next synchronisation (PPO_Agent: synchronously after the
//     first rollout -> falls back to per-step launches and redoes it; afterwards at the read-back of every update -> raises).
#include "common.h"
#include "rng.h"
#include "cartpole.h"
#include "mlp_tile.h"

namespace xrl {

constexpr int QH = 128, QLD = QH + 4;
constexpr int PB_FLAGS = 64, PB_MASK = 96;        // start-up barrier flags / XCC mask of the launch inside the barrier scratch
constexpr int QI_W0 = 0, QI_B0 = 4 * QH, QI_BM = QI_B0 + QH, QI_WH = QI_BM + 2 * QH, QI_LDH = 2 * QH + 4, QI_BH = QI_WH + 3 * QI_LDH;

template <int CTRL>
__device__ __forceinline__ float qdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double qdpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float q_row16_sum(float v) {
    v += qdpp<0x128>(v); v += qdpp<0x124>(v); v += qdpp<0x122>(v); v += qdpp<0x121>(v);
    return v;
}
__device__ __forceinline__ float q_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// device-scope loads: served by L2, never by this CU's L1 (which may hold the previous step's line)
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_dev_u8(const uint8_t* p) {
    // byte flags: read the enclosing aligned word with a device-scope load
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const unsigned w = __hip_atomic_load(reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)((w >> (8 * (a & 3))) & 0xffu);
}
__device__ __forceinline__ float4 ld_dev4(const float* p) { return make_float4(ld_dev(p), ld_dev(p + 1), ld_dev(p + 2), ld_dev(p + 3)); }
// device-scope stores: for the exchange between workgroups that do NOT share an L2 (see `multi` below)
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev_u8(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev4(float* p, float a, float b, float c, float d) { st_dev(p, a); st_dev(p + 1, b); st_dev(p + 2, c); st_dev(p + 3, d); }

template <int ACT, int NJ>
__global__ void __launch_bounds__(FUSED_THREADS) rollout_persistent_kernel(xrl_rollout_persist_t q) {
#pragma clang fp contract(off)
    if (blockIdx.x & 7) return;                                  // keep one XCD's share of the grid (see header)
    __shared__ __attribute__((aligned(16))) float h1[FT * QLD];
    __shared__ __attribute__((aligned(16))) float h2[FT * QLD];
    __shared__ double part[2 * NW * 4];
    __shared__ double ph_state[2][FT][4];
    __shared__ int ph_term[2][FT];
    __shared__ double rs_state[FT][4];
    __shared__ double cp_lds[FT][4];                             // simulator state of this tile's envs
    __shared__ int ep_lds[FT];                                   // episode counters
    __shared__ float s_u[FT];
    __shared__ float s_ret[2];
    __shared__ float s_norm[8];
    __shared__ int s_abort, s_multi;

    const xrl_rollout_step_t& p = q.step0;
    constexpr int D = 4, A = 2;
    const int tid = threadIdx.x, n = p.n, T = q.T;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x >> 3, n_wg = gridDim.x >> 3;
    const int n_act_tiles = (n + FT - 1) / FT;
    const int role = wg / n_act_tiles, tile = wg - role * n_act_tiles;   // 0 act/actor, 1 act/critic, 2 bootstrap/critic
    const bool actor = role == 0, boot = role == 2;
    const bool use_norm = p.use_obsnorm != 0;
    const int e0 = tile * FT;
    const float* img = p.cache_image;
    const int cbase = actor ? 0 : QH;
    const bool mat = wave < 4;
    const int vt = tid - 256, vr = vt >> 3, vs = vt & 7;
    const int r = tid >> 4, sub = tid & 15, e_row = e0 + r;
    const bool row_ok = e_row < n;
    const bool tail_lane = sub == 0 && row_ok;
    const int eh = e0 + li;

    if (tid == 0) {
        s_abort = 0; s_multi = 0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (wg == 0) __hip_atomic_store(q.status + 1, (int)(xcc & 0xf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(q.status + 2, 1 << (xcc & 0xf));                                 // every XCC ever seen (diagnostics, sticky)
        const unsigned seen = atomicOr(q.barrier + PB_MASK, 1u << (xcc & 0xf));   // the XCCs of THIS launch (scratch is zeroed per call)
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");                // performed before the flag below is published
        __hip_atomic_store(q.barrier + PB_FLAGS + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ================= one-time loads: parameters into registers, simulator state into registers / LDS =================
    float4 big[PD];                                          // matrix waves: B fragments; vector waves: first-layer weights
    float4 wh[A][2], b0r[4];
    float bh[A];
    float bm = 0.f;
    if (mat) {
        const float4* fr = reinterpret_cast<const float4*>(p.frag_image) + (size_t)(cbase / 32 + wave) * (QH / 8) * 64 + lane;
#pragma unroll
        for (int c = 0; c < PD; ++c) big[c] = fr[c * 64];
        bm = img[QI_BM + cbase + wave * 32 + li];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) big[j] = *reinterpret_cast<const float4*>(img + QI_W0 + (vs * 16 + j) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0r[j] = *reinterpret_cast<const float4*>(img + QI_B0 + vs * 16 + 4 * j);
    }
    if (actor) {
#pragma unroll
        for (int c = 0; c < A; ++c) {
            wh[c][0] = *reinterpret_cast<const float4*>(img + QI_WH + c * QI_LDH + 4 * sub);
            wh[c][1] = *reinterpret_cast<const float4*>(img + QI_WH + c * QI_LDH + 64 + 4 * sub);
            bh[c] = img[QI_BH + c];
        }
    } else {
        wh[0][0] = *reinterpret_cast<const float4*>(img + QI_WH + A * QI_LDH + QH + 4 * sub);
        wh[0][1] = *reinterpret_cast<const float4*>(img + QI_WH + A * QI_LDH + QH + 64 + 4 * sub);
        bh[0] = img[QI_BH + A];
        wh[1][0] = wh[1][1] = make_float4(0.f, 0.f, 0.f, 0.f); bh[1] = 0.f;
    }
    // per-env state owned by the actor workgroup's tail lanes
    int cp_steps = 0, cp_ep = 0;
    float cp_score = 0.f, rtrack = 0.f;
    if (actor && tail_lane) {
        cp_steps = p.cp_steps[e_row]; cp_score = p.cp_score[e_row]; rtrack = p.ret_track[e_row]; cp_ep = p.cp_episodes[e_row];
        cp_lds[r][0] = p.cp_state[(size_t)e_row * 4 + 0]; cp_lds[r][1] = p.cp_state[(size_t)e_row * 4 + 1];
        cp_lds[r][2] = p.cp_state[(size_t)e_row * 4 + 2]; cp_lds[r][3] = p.cp_state[(size_t)e_row * 4 + 3];
        ep_lds[r] = cp_ep;
    }
    // running statistics, carried in registers (identical in every workgroup)
    float st_mean = 0.f, st_var = 1.f;
    double st_cnt = 0.0;
    if (!mat && lane < D && use_norm) { st_mean = p.obs_stats_in[lane]; st_var = p.obs_stats_in[D + lane]; st_cnt = *p.obs_count_in; }
    float ret_mean = 0.f, ret_var = 1.f;
    double ret_cnt = 0.0;
    if (role == 1 && wave == 6) { ret_mean = p.ret_stats_in[0]; ret_var = p.ret_stats_in[1]; ret_cnt = *p.ret_count_in; }
    const uint32_t step_base = p.step + (p.step_dev ? *p.step_dev : 0u);
    // ================= placement check: do the workgroups of this launch share ONE L2? =================
    // Every workgroup has published its XCC id; once all flags are up the mask is complete.  One bit: the hand-off through
    // plain stores + device-scope loads holds (the case the launch geometry aims for).  More bits (a partitioned device,
    // another process sharing the GPU, a different dispatch order): the exchange stores become device-scope atomic stores,
    // which reach the coherence point of the whole device -- slower per step, same results.  status[3] counts such launches.
    if (wave == 0) {
        int spins = 0;
        for (;;) {
            const unsigned f = lane < n_wg ? __hip_atomic_load(q.barrier + PB_FLAGS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            if (__ballot(f == 0u) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023) == 0 && (spins > 4000000 || __hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (lane == 0) { __hip_atomic_store(q.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; }
                break;
            }
        }
        if (lane == 0) {
            const unsigned mask = __hip_atomic_load(q.barrier + PB_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int m = (__popc(mask) != 1 || (q.flags & 1)) ? 1 : 0;
            s_multi = m;
            if (m && wg == 0) atomicAdd(q.status + 3, 1);
        }
    }
    __syncthreads();
    const bool multi = s_multi != 0;
    if (s_abort) return;                                          // time-out before anything was touched

    for (int t = 0; t <= T; ++t) {
        const bool odd = (t & 1) != 0;
        const bool active = boot ? (t >= 1) : (t < T);
        // ping-pong selection (the host passes buffer 0 as *_in and buffer 1 as *_out of step 0)
        const float* obs_raw_in = odd ? p.obs_raw_out : p.obs_raw_in;   float* obs_raw_out = odd ? const_cast<float*>(p.obs_raw_in) : p.obs_raw_out;
        const float* xnext_in = odd ? p.xnext_out : p.xnext_in;         float* xnext_out = odd ? const_cast<float*>(p.xnext_in) : p.xnext_out;
        const uint8_t* ended_in = odd ? p.ended_out : p.ended_in;       uint8_t* ended_out = odd ? const_cast<uint8_t*>(p.ended_in) : p.ended_out;
        const float* ret_final_in = odd ? p.ret_final_out : p.ret_final_in; float* ret_final_out = odd ? const_cast<float*>(p.ret_final_in) : p.ret_final_out;
        const bool last_step = t == T - 1;

        if (active) {
            if (mat) {
                // -------------------------------------------------------------- matrix waves
                if (use_norm && !boot) lds_barrier();                                              // #1 (statistics)
                lds_barrier();                                                                     // #2 (h1 ready)
                const float* arow = h1 + li * QLD + 4 * lh;
                float4 af[PD];
#pragma unroll
                for (int c = 0; c < PD; ++c) af[c] = *reinterpret_cast<const float4*>(arow + c * 8);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int c = 0; c < PD; ++c) { MFMA4(af[c], big[c], acc) }
                const int col = wave * 32 + li;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                    h2[row * QLD + col] = act_apply_c<ACT>(acc[rr] + bm);
                }
                lds_barrier();                                                                     // #3 (h2 ready)
            } else {
                // -------------------------------------------------------------- vector waves
                constexpr int NS = NJ / 2;
                float sv[2][NS];
                float4 xrow = make_float4(0.f, 0.f, 0.f, 0.f);
                int en[NJ];
                float rfin[NJ];
                if (use_norm && !boot) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < NS; ++j) sv[h][j] = ld_dev(obs_raw_in + min(vt + h * 256 + j * FUSED_THREADS, n * D - 1));
                }
                if (e0 + vr < n) xrow = ld_dev4((boot ? xnext_in : obs_raw_in) + (size_t)(e0 + vr) * D);
                if (role == 1 && wave == 6) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rfin[j] = ld_dev(ret_final_in + min(j * 64 + lane, n - 1));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) en[j] = ld_dev_u8(ended_in + min(j * 64 + lane, n - 1));
                }
                // ---- obs_rms.update(obs) over ALL envs, redundantly per workgroup and per vector wave
                if (use_norm && !boot) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        double s1 = 0.0, s2 = 0.0;
#pragma unroll
                        for (int j = 0; j < NS; ++j) { const double v = vt + h * 256 + j * FUSED_THREADS < n * D ? (double)sv[h][j] : 0.0; s1 += v; s2 += v * v; }
                        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                        s1 += qdpp<0x128>(s1);  s2 += qdpp<0x128>(s2);
                        s1 += qdpp<0x124>(s1);  s2 += qdpp<0x124>(s2);
                        const int vw = wave - 4 + 4 * h;
                        if (lane < D) { part[vw * 4 + lane] = s1; part[NW * 4 + vw * 4 + lane] = s2; }
                    }
                    lds_barrier();                                                                 // #1
                    float new_sd = 1.f;
                    if (lane < D) {
                        double a = 0.0, b = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) { a += part[w * 4 + lane]; b += part[NW * 4 + w * 4 + lane]; }
                        // n a power of two: scaling by 1/n is the exact same number as the division (and ~400 cycles shorter)
                        const bool pow2 = (n & (n - 1)) == 0;
                        const double inv_n = 1.0 / (double)n;
                        const double m = pow2 ? a * inv_n : a / n;
                        const float bmean = (float)m;
                        const float bstd = (float)sqrt(fmax((pow2 ? b * inv_n : b / n) - m * m, 0.0));
                        const float bv = bstd * bstd;
                        const double cnt = st_cnt, tot = cnt + (double)n;
                        const float delta = bmean - st_mean;
                        const float new_mean = st_mean + delta * (float)n / (float)tot;
                        const float m_a = st_var * (float)cnt, m_b = bv * (float)n;
                        const float M2 = m_a + m_b + (delta * delta) * (float)cnt * (float)n / (float)tot;
                        const float new_var = M2 / (float)tot;
                        new_sd = sqrtf(new_var);
                        st_mean = new_mean; st_var = new_var; st_cnt = tot;
                        if (wave == 4) {
                            s_norm[lane] = new_mean; s_norm[4 + lane] = new_sd;
                            if (actor && tile == 0 && last_step) {          // final statistics -> the buffer the next launch reads
                                float* so = (T & 1) ? p.obs_stats_out : const_cast<float*>(p.obs_stats_in);
                                double* co = (T & 1) ? p.obs_count_out : const_cast<double*>(p.obs_count_in);
                                so[lane] = new_mean; so[D + lane] = new_var;
                                if (lane == 0) *co = tot;
                            }
                        }
                    }
                    float nm[4], nsd[4];
#pragma unroll
                    for (int d = 0; d < D; ++d) { nm[d] = q_bcast(st_mean, d); nsd[d] = q_bcast(new_sd, d); }
                    xrow.x = fminf(fmaxf((xrow.x - nm[0]) / (nsd[0] + 1e-8f), -p.obs_range), p.obs_range);
                    xrow.y = fminf(fmaxf((xrow.y - nm[1]) / (nsd[1] + 1e-8f), -p.obs_range), p.obs_range);
                    xrow.z = fminf(fmaxf((xrow.z - nm[2]) / (nsd[2] + 1e-8f), -p.obs_range), p.obs_range);
                    xrow.w = fminf(fmaxf((xrow.w - nm[3]) / (nsd[3] + 1e-8f), -p.obs_range), p.obs_range);
                }
                // ---- first layer on the VALU
                if (actor && vs == 0 && e0 + vr < n)
                    *reinterpret_cast<float4*>(p.obs_slot + ((size_t)t * n + e0 + vr) * D) = xrow;   // memory.observations[t]
                {
                    float* dst = h1 + vr * QLD + vs * 16;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float bq[4] = {b0r[g].x, b0r[g].y, b0r[g].z, b0r[g].w};
                        float o[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const float4 w = big[g * 4 + jj];
                            float acc = __fmaf_rn(xrow.x, w.x, 0.f);
                            acc = __fmaf_rn(xrow.y, w.y, acc);
                            acc = __fmaf_rn(xrow.z, w.z, acc);
                            acc = __fmaf_rn(xrow.w, w.w, acc);
                            o[jj] = act_apply_c<ACT>(acc + bq[jj]);
                        }
                        *reinterpret_cast<float4*>(dst + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
                lds_barrier();                                                                     // #2
                // ---- while the matrix cores run: everything of the tail that does not depend on the logits
                if (actor) {
                    if (wave == 4) {                         // envs.step for both actions: lanes 0-31 a = 0, lanes 32-63 a = 1
                        double cps[4] = {cp_lds[li][0], cp_lds[li][1], cp_lds[li][2], cp_lds[li][3]};
                        double x, xd, th, thd;
                        bool term;
                        cartpole_advance(cps, lh, x, xd, th, thd, term);
                        ph_state[lh][li][0] = x; ph_state[lh][li][1] = xd; ph_state[lh][li][2] = th; ph_state[lh][li][3] = thd;
                        ph_term[lh][li] = term ? 1 : 0;
                    } else if (wave == 5) {                  // state after an auto-reset into the next episode
                        uint32_t o[4], qq[4];
                        philox4x32(p.env_seed, (uint32_t)eh, (uint32_t)(ep_lds[li] + 1), lh ? STREAM_RESET_B : STREAM_RESET_A, o);
#pragma unroll
                        for (int j = 0; j < 4; ++j) qq[j] = __shfl_xor(o[j], 32, 64);
                        if (lh == 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) rs_state[li][j] = -0.05 + 0.1 * u01d(o[j], qq[j]);
                        }
                    } else if (wave == 6) {                  // sampling uniform of (env, step)
                        uint32_t rr4[4];
                        philox4x32(p.seed, (uint32_t)eh, step_base + (uint32_t)t, STREAM_ACTION, rr4);
                        if (lh == 0) s_u[li] = u01(rr4[0]);
                    }
                } else if (role == 1 && wave == 6) {
                    // deferred ret_rms.update() of the episodes that ended at the previous step, in env order
                    float mean = ret_mean, var = ret_var;
                    double count = ret_cnt;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        unsigned long long mm = __ballot(j * 64 + lane < n && en[j] != 0);
                        while (mm) {
                            const int bpos = __ffsll((long long)mm) - 1; mm &= mm - 1;
                            const float bmv = __shfl(rfin[j], bpos, 64);
                            const double tot = count + 1.0; const float delta = bmv - mean;
                            const float new_mean = mean + delta * 1.0f / (float)tot;
                            const float M2 = var * (float)count + 0.f + (delta * delta) * (float)count * 1.0f / (float)tot;
                            mean = new_mean; var = M2 / (float)tot; count = tot;
                        }
                    }
                    ret_mean = mean; ret_var = var; ret_cnt = count;
                    if (lane == 0) {
                        s_ret[0] = mean; s_ret[1] = var;
                        if (tile == 0 && last_step) {
                            float* ro = (T & 1) ? p.ret_stats_out : const_cast<float*>(p.ret_stats_in);
                            double* co = (T & 1) ? p.ret_count_out : const_cast<double*>(p.ret_count_in);
                            ro[0] = mean; ro[1] = var; *co = count;
                        }
                    }
                }
                lds_barrier();                                                                     // #3
            }

            // ================= heads on the VALU: 16 threads per row =================
            float hv[A];
            {
                const float4 a0 = *reinterpret_cast<const float4*>(h2 + r * QLD + 4 * sub);
                const float4 a1 = *reinterpret_cast<const float4*>(h2 + r * QLD + 64 + 4 * sub);
#pragma unroll
                for (int c = 0; c < A; ++c) {
                    float acc = 0.f;
                    acc += a0.x * wh[c][0].x + a0.y * wh[c][0].y + a0.z * wh[c][0].z + a0.w * wh[c][0].w;
                    acc += a1.x * wh[c][1].x + a1.y * wh[c][1].y + a1.z * wh[c][1].z + a1.w * wh[c][1].w;
                    hv[c] = q_row16_sum(acc) + bh[c];
                }
            }
            if (tail_lane) {
                const int e = e_row;
                if (boot) {
                    q.bootv[(size_t)(t - 1) * n + e] = hv[0];                    // V(next_obs_{t-1}) -> bootv[t-1]
                } else if (!actor) {
                    p.val_slot[(size_t)t * n + e] = hv[0];
                    float rstd = sqrtf(s_ret[1]);
                    rstd = fminf(fmaxf(rstd, 0.1f), 100.f);
                    float rn = 1.0f;
                    if (p.use_rewnorm) rn = fminf(fmaxf(1.0f / rstd, -p.rew_range), p.rew_range);
                    p.rew_slot[(size_t)t * n + e] = rn;
                } else {
                    // ---- get_actions: sample, log-prob; store
                    int a;
                    float logp;
                    {
                        const float u = s_u[r];
                        const float mx = fmaxf(hv[0], hv[1]);
                        float se = 0.f;
                        se += expf(hv[0] - mx); se += expf(hv[1] - mx);
                        const float lse = mx + logf(se);
                        float c = 0.f;
                        c += expf(hv[0] - lse);
                        a = c > u ? 0 : 1;
                        logp = hv[a] - lse;
                    }
                    p.act_slot[(size_t)t * n + e] = (float)a;
                    p.logp_slot[(size_t)t * n + e] = logp;
                    // ---- envs.step(acts): pick the pre-computed transition + auto-reset
                    const double x = ph_state[a][r][0], xd = ph_state[a][r][1], th = ph_state[a][r][2], thd = ph_state[a][r][3];
                    const bool term = ph_term[a][r] != 0;
                    const int steps = cp_steps + 1;
                    const bool trunc = steps >= p.max_steps;
                    const float nobs[4] = {(float)x, (float)xd, (float)th, (float)thd};
                    const float score = cp_score + 1.0f;
                    float robs[4] = {nobs[0], nobs[1], nobs[2], nobs[3]};
                    if (term || trunc) {
                        cp_ep += 1;
                        const double r0 = rs_state[r][0], r1 = rs_state[r][1], r2 = rs_state[r][2], r3 = rs_state[r][3];
                        cp_lds[r][0] = r0; cp_lds[r][1] = r1; cp_lds[r][2] = r2; cp_lds[r][3] = r3;
                        ep_lds[r] = cp_ep;
                        cp_steps = 0; cp_score = 0.f;
                        robs[0] = (float)r0; robs[1] = (float)r1; robs[2] = (float)r2; robs[3] = (float)r3;
                        atomicAdd(&p.cp_stats[0], 1.0); atomicAdd(&p.cp_stats[1], (double)score); atomicAdd(&p.cp_stats[2], (double)steps);
                    } else {
                        cp_lds[r][0] = x; cp_lds[r][1] = xd; cp_lds[r][2] = th; cp_lds[r][3] = thd;
                        cp_steps = steps; cp_score = score;
                    }
                    // ---- bookkeeping
                    const float reward = 1.0f;
                    p.term_slot[(size_t)t * n + e] = term ? 1.f : 0.f;
                    p.seg_slot[(size_t)t * n + e] = (term || trunc || last_step) ? (uint8_t)(1 | (term ? 6 : 0)) : (uint8_t)0;
                    const float tr = p.gamma * rtrack + reward;
                    const bool fin = term || trunc;
                    if (multi) {
                        if (fin) st_dev(ret_final_out + e, tr);
                        st_dev_u8(ended_out + e, fin ? 1 : 0);
                        st_dev4(obs_raw_out + (size_t)e * 4, robs[0], robs[1], robs[2], robs[3]);
                    } else {
                        if (fin) ret_final_out[e] = tr;
                        ended_out[e] = fin ? 1 : 0;
                        *reinterpret_cast<float4*>(obs_raw_out + (size_t)e * 4) = make_float4(robs[0], robs[1], robs[2], robs[3]);
                    }
                    rtrack = fin ? 0.f : tr;
                    float nv[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float v = nobs[d];
                        if (p.use_obsnorm) { v = (v - s_norm[d]) / (s_norm[4 + d] + 1e-8f); v = fminf(fmaxf(v, -p.obs_range), p.obs_range); }
                        nv[d] = v;
                    }
                    if (multi) st_dev4(xnext_out + (size_t)e * 4, nv[0], nv[1], nv[2], nv[3]);
                    else *reinterpret_cast<float4*>(xnext_out + (size_t)e * 4) = make_float4(nv[0], nv[1], nv[2], nv[3]);
                }
            }
        }

        // ================= step boundary: counter barrier across the workgroups =================
        if (t < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's stores have reached L2
            __syncthreads();
            // flag barrier: workgroup w publishes barrier[w] = t + 1 with a plain device-scope store; wave 0 polls all flags
            // with ONE coalesced load per round (no read-modify-write atomics serialising in L2)
            if (tid == 0) __hip_atomic_store(q.barrier + wg, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wave == 0) {
                int spins = 0;
                for (;;) {
                    const unsigned f = lane < n_wg ? __hip_atomic_load(q.barrier + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                   : 0xffffffffu;
                    if (__ballot(f < (unsigned)(t + 1)) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023) == 0) {
                        if (spins > 4000000 || __hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            if (lane == 0) { __hip_atomic_store(q.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; }
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            if (s_abort) break;
        }
    }

    // ================= hand the simulator state back =================
    if (actor && tail_lane) {
        p.cp_steps[e_row] = cp_steps; p.cp_score[e_row] = cp_score; p.ret_track[e_row] = rtrack; p.cp_episodes[e_row] = cp_ep;
        p.cp_state[(size_t)e_row * 4 + 0] = cp_lds[r][0]; p.cp_state[(size_t)e_row * 4 + 1] = cp_lds[r][1];
        p.cp_state[(size_t)e_row * 4 + 2] = cp_lds[r][2]; p.cp_state[(size_t)e_row * 4 + 3] = cp_lds[r][3];
    }
}

__global__ void zero_words_kernel(uint32_t* p) { p[threadIdx.x] = 0u; }

bool rollout_fast_eligible(const xrl_rollout_step_t& p);

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_rollout_cartpole_persistent(const xrl_rollout_persist_t* qq, xrl_stream_t stream) {
    XRL_CHECK_ARG(qq != nullptr);
    const xrl_rollout_persist_t& q = *qq;
    const xrl_rollout_step_t& p = q.step0;
    XRL_CHECK_ARG(q.T >= 1 && q.bootv && q.barrier && q.status);
    XRL_CHECK_ARG(p.params && p.n > 0 && p.cache_image && p.frag_image);
    XRL_CHECK_ARG(p.obs_raw_in && p.obs_raw_out && p.xnext_in && p.xnext_out && p.obs_stats_in && p.obs_stats_out && p.obs_count_in &&
                  p.obs_count_out && p.ret_stats_in && p.ret_stats_out && p.ret_count_in && p.ret_count_out && p.ended_in &&
                  p.ended_out && p.ret_final_in && p.ret_final_out && p.ret_track);
    XRL_CHECK_ARG(p.obs_slot && p.act_slot && p.val_slot && p.logp_slot && p.rew_slot && p.term_slot && p.seg_slot);
    XRL_CHECK_ARG(p.cp_state && p.cp_steps && p.cp_episodes && p.cp_score && p.cp_stats);
    XRL_CHECK_ARG(((reinterpret_cast<uintptr_t>(p.ended_in) | reinterpret_cast<uintptr_t>(p.ended_out)) & 3) == 0);
    if (!rollout_fast_eligible(p)) { set_error("xrl_rollout_cartpole_persistent: network is not the 4-128-{128-2,128-1} class"); return XRL_EINVAL; }
    const int n_tiles = (p.n + FT - 1) / FT;
    const int n_wg = 3 * n_tiles;
    XRL_CHECK_ARG(n_wg <= device_cu_count() / 8);                   // all resident workgroups on ONE XCD, one per CU
    // The barrier scratch is zeroed by a KERNEL, not by hipMemsetAsync: inside a captured hipGraph a memset node does not
    // hold back the kernel node behind it until the kernel node in front of it (xrl_pack_rollout_cache) has finished --
    // measured on ROCm 7.2 / MI355X with tools/stress_determinism.py: with the memset node 16 of 29 replays of the rollout
    // graph read partly stale parameter images, with a zeroing kernel 0 of 29 (eager launches were never affected).
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(128), 0, as_stream(stream), q.barrier);
    XRL_ACT_DISPATCH(p.layers[0].act,
        if (p.n <= 256) hipLaunchKernelGGL((rollout_persistent_kernel<ACT, 4>), dim3(8 * n_wg), dim3(FUSED_THREADS), 0, as_stream(stream), q);
        else hipLaunchKernelGGL((rollout_persistent_kernel<ACT, 16>), dim3(8 * n_wg), dim3(FUSED_THREADS), 0, as_stream(stream), q);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
