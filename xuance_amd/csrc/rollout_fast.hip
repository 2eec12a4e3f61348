// Shape-specialised fused rollout step for the CartPole actor-critic 4-128-{128-2, 128-1} (BASELINE configs[0..1]):
// same dataflow contract, same arithmetic and bit-identical results as rollout_step_cartpole_kernel
// (rollout_fused.hip, the any-shape kernel), but every extent is a compile-time constant.
//
// Why a second kernel (measured with the shader-clock stamps, tools/microbench_rollout.py): the any-shape kernel is
// 139 KB of code whose run-time layer tables leave hipcc with un-unrollable loops -- every LDS access becomes its own
// latency (8 waves per CU cannot hide them), every `q < kq` a branch.  Its actor workgroup needs 48 k cycles for a step
// whose matrix-core work is 4 k cycles.  Here:
//   * NO LDS parameter cache: first-layer weights, biases and head weights go from the packed image straight into the
//     registers of the threads that use them, the 128x128 branch weights into B-fragment registers (waves 0-3);
//     everything is issued before the first wait, in the order it is needed (vmcnt retires in order);
//   * the observation statistics are finalised redundantly by every wave (no second barrier), 16-lane reductions are
//     DPP row rotations (bit-identical to the xor butterfly, see dpp_ror_sum16);
//   * while waves 0-3 run the 64 chained MFMAs of the branch layer, the otherwise idle waves 4-6 of the actor workgroup
//     pre-compute everything that does not depend on the logits: the physics step for BOTH actions, the reset state of
//     the next episode, the sampling uniform and the deferred return-statistics merge;
//   * the tail runs in the 32 lanes that already hold their row's logits (no head level in LDS, no fourth barrier).
// Reference semantics as in rollout_fused.hip (ppo_agent.py:113-177, core/on_policy.py:109-169).
#include "common.h"
#include "rng.h"
#include "cartpole.h"
#include "mlp_tile.h"

namespace xrl {

constexpr int RH = 128;                      // hidden width of the shape class
constexpr int RLD = RH + 4;                  // LDS row stride (conflict-free ds_read_b128 of A fragments)
// packed image layout (pack_rollout_cache_kernel) for 4-128-256-{2|1}
constexpr int IMG_W0 = 0, IMG_B0 = 4 * RH, IMG_BM = IMG_B0 + RH, IMG_WH = IMG_BM + 2 * RH, IMG_LDH = 2 * RH + 4,
              IMG_BH = IMG_WH + 3 * IMG_LDH;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROR1 = 0x121, DPP_ROR2 = 0x122, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;
// v[i] + v[i^8], then ^4, ^2, ^1 inside each row of 16 lanes.  A rotation by 8 IS the xor-8 exchange; after it the
// values have period 8 inside the row, so rotating by 4 reads the same number as lane i^4 would supply, and so on:
// bit-identical to the __shfl_xor butterfly of narrow_layer_valu, without the LDS crossbar.
__device__ __forceinline__ float dpp_ror_sum16(float v) {
    v += dpp_mov<DPP_ROR8>(v); v += dpp_mov<DPP_ROR4>(v); v += dpp_mov<DPP_ROR2>(v); v += dpp_mov<DPP_ROR1>(v);
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// NJ = ceil(n_envs / 64) rounded up to 4 or 16: how many per-lane words the all-env arrays need (register budget).
template <int ACT, int NJ>
__global__ void __launch_bounds__(FUSED_THREADS) rollout_step_fast_kernel(xrl_rollout_step_t p) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float h1[FT * RLD];
    __shared__ __attribute__((aligned(16))) float h2[FT * RLD];
    __shared__ double part[2 * NW * 4];
    __shared__ double ph_state[2][FT][4];
    __shared__ int ph_term[2][FT];
    __shared__ double rs_state[FT][4];
    __shared__ float s_u[FT];
    __shared__ float s_ret[2];
    __shared__ float s_norm[8];                             // obs mean[4] | std[4] after this step's update

    kernarg_prefetch<sizeof(xrl_rollout_step_t)>();
    constexpr int D = 4, A = 2;
    const int tid = threadIdx.x, n = p.n;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-role branches become s_cbranch, no exec juggling
    long long* dbg = p.dbg;
    // diagnostics: shader-clock stamps are collected in registers and written at the very end (a store issued in the
    // middle of the kernel queues behind the weight stream and would stall the stamped wave)
    const bool dbg_me = dbg && tid == 256 && (int)blockIdx.x == (int)dbg[63];      // stamps follow a vector wave
    const bool dbg_wave = dbg && lane == 0 && (int)blockIdx.x == (int)dbg[63];
    long long tst[14], tw0 = 0, tw1 = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) tst[i] = 0;
#define FSTAMP(k) do { if (dbg_me) tst[k] = clock64(); } while (0)
#define FLUSH_STAMPS() do { if (dbg_me) { int c_ = 0; _Pragma("unroll") for (int i = 0; i < 14; ++i) if (tst[i]) dbg[c_++] = tst[i]; dbg[15] = c_; } \
                           if (dbg_wave) { dbg[16 + wave * 2] = tw0; dbg[17 + wave * 2] = tw1; } } while (0)
    FSTAMP(0);
    const int n_act_tiles = (n + FT - 1) / FT;
    const int group = blockIdx.x / n_act_tiles, tile = blockIdx.x - group * n_act_tiles;
    const int role = p.boot_only ? 2 : group;              // 0 act/actor, 1 act/critic, 2 bootstrap/critic
    const bool actor = role == 0, boot = role == 2;
    const bool use_norm = !boot && p.use_obsnorm;
    const int e0 = tile * FT;
    const float* img = p.cache_image;
    const int cbase = actor ? 0 : RH;
    // waves 0-3 ("matrix waves") own the 64 KB B-fragment stream and the MFMAs; waves 4-7 ("vector waves") own everything
    // that consumes small loads early (statistics, normalisation, first layer).  vmcnt retires in order, so a wave that has
    // the weight stream in flight cannot wait for a small load without waiting for the stream.
    const bool mat = wave < 4;
    const int vt = tid - 256;                               // vector-thread index 0..255 (negative on matrix waves)
    const int vr = vt >> 3, vs = vt & 7;                    // first layer: row vr, columns [16 vs, 16 vs + 16)
    const int r = tid >> 4, sub = tid & 15, e_row = e0 + r; // heads / tail: 16 threads per row
    const bool row_ok = e_row < n;
    const bool tail_lane = sub == 0 && row_ok;

    // ================= every global load of the launch =================
    // The two kinds of waves run two separate straight-line programs that meet at the barriers (same barrier count on
    // both sides): the instruction cache is cold at every launch and each far branch target costs ~1 k cycles, so neither
    // program hops over the other's blocks.
    // Order inside the CU's (in-order) vector memory pipeline: all small loads of all waves first, then the weight stream:
    // the vector waves issue theirs and arrive at a bare s_barrier; the matrix waves start the stream behind it.
    float4 big[PD];                                         // matrix waves: B fragments; vector waves: first-layer weights
    float4 wh[A][2];
    float bh[A];
    int cp_steps0 = 0, cp_ep0 = 0;
    float cp_score0 = 0.f, rtrack0 = 0.f;
    // heads: chunks q = sub, sub + 16 of this workgroup's half of the merged head rows; tail inputs on lanes sub == 0
#define LATE_LOADS()                                                                                                   \
    do {                                                                                                               \
        if (actor) {                                                                                                   \
            _Pragma("unroll") for (int c = 0; c < A; ++c) {                                                            \
                wh[c][0] = *reinterpret_cast<const float4*>(img + IMG_WH + c * IMG_LDH + 4 * sub);                     \
                wh[c][1] = *reinterpret_cast<const float4*>(img + IMG_WH + c * IMG_LDH + 64 + 4 * sub);                \
                bh[c] = img[IMG_BH + c];                                                                               \
            }                                                                                                          \
            if (tail_lane) {                                                                                           \
                cp_steps0 = p.cp_steps[e_row]; cp_score0 = p.cp_score[e_row]; rtrack0 = p.ret_track[e_row];            \
                cp_ep0 = p.cp_episodes[e_row];                                                                         \
            }                                                                                                          \
        } else {                                                                                                       \
            wh[0][0] = *reinterpret_cast<const float4*>(img + IMG_WH + A * IMG_LDH + RH + 4 * sub);                    \
            wh[0][1] = *reinterpret_cast<const float4*>(img + IMG_WH + A * IMG_LDH + RH + 64 + 4 * sub);               \
            bh[0] = img[IMG_BH + A];                                                                                   \
            wh[1][0] = wh[1][1] = make_float4(0.f, 0.f, 0.f, 0.f); bh[1] = 0.f;                                        \
        }                                                                                                              \
    } while (0)

    if (mat) {
        // ------------------------------------------------------------------ matrix waves
        asm volatile("s_barrier" ::: "memory");                                                    // #0
        const float4* fr = reinterpret_cast<const float4*>(p.frag_image) + (size_t)(cbase / 32 + wave) * (RH / 8) * 64 + lane;
#pragma unroll
        for (int q = 0; q < PD; ++q) big[q] = fr[q * 64];
        const float bm = img[IMG_BM + cbase + wave * 32 + li];
        LATE_LOADS();
        if (use_norm) lds_barrier();                                                               // #1 (statistics)
        lds_barrier();                                                                             // #2 (h1 ready)
        if (dbg_wave) tw0 = clock64();
        const float* arow = h1 + li * RLD + 4 * lh;
        float4 af[PD];
#pragma unroll
        for (int q = 0; q < PD; ++q) af[q] = *reinterpret_cast<const float4*>(arow + q * 8);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; q < PD; ++q) { MFMA4(af[q], big[q], acc) }
        const int col = wave * 32 + li;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h2[row * RLD + col] = act_apply_c<ACT>(acc[rr] + bm);
        }
        if (dbg_wave) tw1 = clock64();
        lds_barrier();                                                                             // #3 (h2 ready)
    } else {
        // ------------------------------------------------------------------ vector waves
        constexpr int NS = NJ / 2;                          // sv words per virtual thread
        float sv[2][NS];
        float st_mean = 0.f, st_var = 1.f;
        double st_cnt = 0.0;
        float4 xrow = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 b0r[4];
        double cps[4] = {0.0, 0.0, 0.0, 0.0};
        int ep_h = 0;
        uint32_t step_dev = 0u;
        int en[NJ];
        float rfin[NJ];
        float ret_m0 = 0.f, ret_v0 = 1.f;
        double ret_c0 = 0.0;
        const int eh = e0 + li;                             // env of a helper-wave lane
        if (use_norm) {                                     // vector thread vt stands in for threads vt and vt + 256 of the
#pragma unroll                                              // any-shape kernel's 512-thread sum (same association order)
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < NS; ++j) sv[h][j] = p.obs_raw_in[min(vt + h * 256 + j * FUSED_THREADS, n * D - 1)];   // clamped: masked at use
            if (lane < D) { st_mean = p.obs_stats_in[lane]; st_var = p.obs_stats_in[D + lane]; st_cnt = *p.obs_count_in; }
        }
        if (e0 + vr < n) xrow = *reinterpret_cast<const float4*>((boot ? p.xnext_in : p.obs_raw_in) + (size_t)(e0 + vr) * D);
#pragma unroll
        for (int j = 0; j < 16; ++j) big[j] = *reinterpret_cast<const float4*>(img + IMG_W0 + (vs * 16 + j) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0r[j] = *reinterpret_cast<const float4*>(img + IMG_B0 + vs * 16 + 4 * j);
        LATE_LOADS();
        if (actor) {                                        // inputs of the helper waves 4-6
            if (wave == 4 && eh < n) {
                cps[0] = p.cp_state[(size_t)eh * 4 + 0]; cps[1] = p.cp_state[(size_t)eh * 4 + 1];
                cps[2] = p.cp_state[(size_t)eh * 4 + 2]; cps[3] = p.cp_state[(size_t)eh * 4 + 3];
            }
            if (wave == 5 && eh < n) ep_h = p.cp_episodes[eh];
            if (wave == 6 && p.step_dev) step_dev = *p.step_dev;
        } else if (role == 1 && wave == 6) {                // the act tile's critic workgroup owns the return statistics
#pragma unroll
            for (int j = 0; j < NJ; ++j) rfin[j] = p.ret_final_in[min(j * 64 + lane, n - 1)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) en[j] = (int)p.ended_in[min(j * 64 + lane, n - 1)];   // unconditional, consumed below
            ret_m0 = p.ret_stats_in[0]; ret_v0 = p.ret_stats_in[1]; ret_c0 = *p.ret_count_in;
        }
        asm volatile("s_barrier" ::: "memory");                                                    // #0

        // ---- obs_rms.update(obs) over ALL envs, redundantly per workgroup and per vector wave
        if (use_norm) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                    // virtual wave (wave - 4) + 4 h of the 512-thread sum
                double s1 = 0.0, s2 = 0.0;                   // dimension d = lane & 3
#pragma unroll
                for (int j = 0; j < NS; ++j) { const double v = vt + h * 256 + j * FUSED_THREADS < n * D ? (double)sv[h][j] : 0.0; s1 += v; s2 += v * v; }
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += dpp_mov<DPP_ROR8>(s1);  s2 += dpp_mov<DPP_ROR8>(s2);
                s1 += dpp_mov<DPP_ROR4>(s1);  s2 += dpp_mov<DPP_ROR4>(s2);
                const int vw = wave - 4 + 4 * h;
                if (lane < D) { part[vw * 4 + lane] = s1; part[NW * 4 + vw * 4 + lane] = s2; }
            }
            FSTAMP(1);
            lds_barrier();                                                                         // #1
            FSTAMP(2);
            float new_mean = 0.f, new_sd = 1.f;
            if (lane < D) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) { a += part[w * 4 + lane]; b += part[NW * 4 + w * 4 + lane]; }
                // n a power of two: scaling by 1/n is the exact same number as the division (and ~400 cycles shorter)
                const bool pow2 = (n & (n - 1)) == 0;
                const double inv_n = 1.0 / (double)n;
                const double m = pow2 ? a * inv_n : a / n;
                const float bmean = (float)m;                 // np.mean -> float32
                const float bstd = (float)sqrt(fmax((pow2 ? b * inv_n : b / n) - m * m, 0.0));
                const float bv = bstd * bstd;                 // batch_var = np.square(batch_std)
                const double cnt = st_cnt, tot = cnt + (double)n;
                const float delta = bmean - st_mean;          // update_from_moments (statistic_tools.py:173-185)
                new_mean = st_mean + delta * (float)n / (float)tot;
                const float m_a = st_var * (float)cnt, m_b = bv * (float)n;
                const float M2 = m_a + m_b + (delta * delta) * (float)cnt * (float)n / (float)tot;
                const float new_var = M2 / (float)tot;
                new_sd = sqrtf(new_var);
                if (wave == 4) {
                    s_norm[lane] = new_mean; s_norm[4 + lane] = new_sd;
                    if (actor && tile == 0) {
                        p.obs_stats_out[lane] = new_mean; p.obs_stats_out[D + lane] = new_var;
                        if (lane == 0) *p.obs_count_out = tot;
                    }
                }
            }
            FSTAMP(3);
            float nm[4], nsd[4];
#pragma unroll
            for (int d = 0; d < D; ++d) { nm[d] = lane_bcast(new_mean, d); nsd[d] = lane_bcast(new_sd, d); }
            xrow.x = fminf(fmaxf((xrow.x - nm[0]) / (nsd[0] + 1e-8f), -p.obs_range), p.obs_range);
            xrow.y = fminf(fmaxf((xrow.y - nm[1]) / (nsd[1] + 1e-8f), -p.obs_range), p.obs_range);
            xrow.z = fminf(fmaxf((xrow.z - nm[2]) / (nsd[2] + 1e-8f), -p.obs_range), p.obs_range);
            xrow.w = fminf(fmaxf((xrow.w - nm[3]) / (nsd[3] + 1e-8f), -p.obs_range), p.obs_range);
        }
        FSTAMP(4);
        // ---- first layer on the VALU (k-ordered fma chain == the MFMA result)
        if (actor && vs == 0 && e0 + vr < n) *reinterpret_cast<float4*>(p.obs_slot + (size_t)(e0 + vr) * D) = xrow;   // memory.observations[t]
        {
            float* dst = h1 + vr * RLD + vs * 16;
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
        FSTAMP(5);
        lds_barrier();                                                                             // #2
        FSTAMP(6);
        // ---- while the matrix cores run: everything of the tail that does not depend on the logits
        if (dbg_wave) tw0 = clock64();
        if (actor) {
            if (wave == 4) {                                 // envs.step for both actions: lanes 0-31 a = 0, lanes 32-63 a = 1
                double x, xd, th, thd;
                bool term;
                cartpole_advance(cps, lh, x, xd, th, thd, term);
                ph_state[lh][li][0] = x; ph_state[lh][li][1] = xd; ph_state[lh][li][2] = th; ph_state[lh][li][3] = thd;
                ph_term[lh][li] = term ? 1 : 0;
            } else if (wave == 5) {                          // state after an auto-reset into episode ep_h + 1 (cartpole_reset):
                uint32_t o[4], q[4];                         // lanes 0-31 draw stream A, lanes 32-63 stream B of the same env
                philox4x32(p.env_seed, (uint32_t)eh, (uint32_t)(ep_h + 1), lh ? STREAM_RESET_B : STREAM_RESET_A, o);
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = __shfl_xor(o[j], 32, 64);
                if (lh == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rs_state[li][j] = -0.05 + 0.1 * u01d(o[j], q[j]);
                }
            } else if (wave == 6) {                          // sampling uniform of (env, step)
                uint32_t rr4[4];
                philox4x32(p.seed, (uint32_t)eh, p.step + step_dev, STREAM_ACTION, rr4);
                if (lh == 0) s_u[li] = u01(rr4[0]);
            }
        } else if (role == 1 && wave == 6) {
            // deferred ret_rms.update() of the episodes that ended at the previous step, in env order (ppo_agent.py:146-149).
            // CartPole's reward is the constant 1, so the normalised reward of this step needs nothing from the actor
            // workgroup: the critic workgroup of the act tile (which has slack) owns the statistics and writes rew_slot.
            float mean = ret_m0, var = ret_v0;
            double count = ret_c0;
            // opaque pass-through: without it hipcc folds `en[j] != 0` into the load section (one full memory round
            // trip per flag word, in front of the first barrier)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(en[j]));
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
            if (lane == 0) {
                s_ret[0] = mean; s_ret[1] = var;
                if (tile == 0) { p.ret_stats_out[0] = mean; p.ret_stats_out[1] = var; *p.ret_count_out = count; }
            }
        }
        if (dbg_wave) tw1 = clock64();
        FSTAMP(7);
        lds_barrier();                                                                             // #3
        FSTAMP(8);
    }
#undef LATE_LOADS

    // ================= heads on the VALU: 16 threads per row, k-slices q = sub, sub + 16 =================
    float hv[A];
    {
        const float4 a0 = *reinterpret_cast<const float4*>(h2 + r * RLD + 4 * sub);
        const float4 a1 = *reinterpret_cast<const float4*>(h2 + r * RLD + 64 + 4 * sub);
#pragma unroll
        for (int c = 0; c < A; ++c) {
            float acc = 0.f;
            acc += a0.x * wh[c][0].x + a0.y * wh[c][0].y + a0.z * wh[c][0].z + a0.w * wh[c][0].w;
            acc += a1.x * wh[c][1].x + a1.y * wh[c][1].y + a1.z * wh[c][1].z + a1.w * wh[c][1].w;
            hv[c] = dpp_ror_sum16(acc) + bh[c];
        }
    }
    FSTAMP(9);
    if (!tail_lane) return;
    const int e = e_row;
    if (!actor) {                                            // hv[0] = V
        if (boot) { p.bootv_prev[e] = hv[0]; FLUSH_STAMPS(); return; }
        p.val_slot[e] = hv[0];
        float rstd = sqrtf(s_ret[1]);                        // reward normalisation with the merged statistics (ppo_agent.py:128)
        rstd = fminf(fmaxf(rstd, 0.1f), 100.f);
        float rn = 1.0f;
        if (p.use_rewnorm) rn = fminf(fmaxf(1.0f / rstd, -p.rew_range), p.rew_range);
        p.rew_slot[e] = rn;
        FLUSH_STAMPS();
        return;
    }
    // ---- get_actions (core/on_policy.py:128-169): sample, log-prob; store (ppo_agent.py:128)
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
        a = c > u ? 0 : 1;                                   // (the last bucket absorbs rounding, as in the any-shape kernel)
        logp = hv[a] - lse;
    }
    FSTAMP(10);
    p.act_slot[e] = (float)a;
    p.logp_slot[e] = logp;
    // ---- envs.step(acts): pick the pre-computed transition + DummyVecEnv auto-reset
    const double x = ph_state[a][r][0], xd = ph_state[a][r][1], th = ph_state[a][r][2], thd = ph_state[a][r][3];
    const bool term = ph_term[a][r] != 0;
    double* s = p.cp_state + (size_t)e * 4;
    const int steps = cp_steps0 + 1;
    const bool trunc = steps >= p.max_steps;
    const float nobs[4] = {(float)x, (float)xd, (float)th, (float)thd};
    const float score = cp_score0 + 1.0f;
    float robs[4] = {nobs[0], nobs[1], nobs[2], nobs[3]};
    if (term || trunc) {
        p.cp_episodes[e] = cp_ep0 + 1;
        const double r0 = rs_state[r][0], r1 = rs_state[r][1], r2 = rs_state[r][2], r3 = rs_state[r][3];
        s[0] = r0; s[1] = r1; s[2] = r2; s[3] = r3;
        p.cp_steps[e] = 0; p.cp_score[e] = 0.f;
        robs[0] = (float)r0; robs[1] = (float)r1; robs[2] = (float)r2; robs[3] = (float)r3;
        atomicAdd(&p.cp_stats[0], 1.0); atomicAdd(&p.cp_stats[1], (double)score); atomicAdd(&p.cp_stats[2], (double)steps);
    } else {
        s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
        p.cp_steps[e] = steps; p.cp_score[e] = score;
    }
    FSTAMP(11);
    // ---- bookkeeping (ppo_agent.py:128,144-157)
    const float reward = 1.0f;                               // (rew_slot: critic workgroup of this tile)
    p.term_slot[e] = term ? 1.f : 0.f;
    p.seg_slot[e] = (term || trunc || p.last_step) ? (uint8_t)(1 | (term ? 6 : 0)) : (uint8_t)0;
    const float tr = p.gamma * rtrack0 + reward;
    if (term || trunc) { p.ret_final_out[e] = tr; p.ended_out[e] = 1; p.ret_track[e] = 0.f; }
    else { p.ended_out[e] = 0; p.ret_track[e] = tr; }
    *reinterpret_cast<float4*>(p.obs_raw_out + (size_t)e * 4) = make_float4(robs[0], robs[1], robs[2], robs[3]);
    float nv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float v = nobs[d];
        if (p.use_obsnorm) { v = (v - s_norm[d]) / (s_norm[4 + d] + 1e-8f); v = fminf(fmaxf(v, -p.obs_range), p.obs_range); }
        nv[d] = v;
    }
    *reinterpret_cast<float4*>(p.xnext_out + (size_t)e * 4) = make_float4(nv[0], nv[1], nv[2], nv[3]);
    FSTAMP(12);
    FLUSH_STAMPS();
#undef FSTAMP
#undef FLUSH_STAMPS
}

static bool g_fast_enabled = true;

// The shape class: D = 4, A = 2, hidden 128 everywhere, one stacked actor|critic branch layer, one activation.
bool rollout_fast_eligible(const xrl_rollout_step_t& p) {
    if (!g_fast_enabled || !p.role_split || !p.frag_image || p.split_col != RH || p.gaussian) return false;
    if (p.D != 4 || p.A != 2 || p.n_layers != 4 || p.n_head_layers != 2 || p.n_levels != 4) return false;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &Ha = p.layers[2], &Hc = p.layers[3];
    if (p.level_width[1] != RH || p.level_width[2] != 2 * RH || p.level_width[3] != 3) return false;
    if (L0.K != 4 || L0.N != RH || L0.in_level != 0 || L0.out_level != 1 || L0.out_off != 0) return false;
    if (L1.K != RH || L1.N != 2 * RH || L1.in_level != 1 || L1.in_off != 0 || L1.out_level != 2 || L1.out_off != 0) return false;
    if (L1.act != L0.act) return false;
    if (Ha.K != RH || Ha.N != 2 || Ha.in_level != 2 || Ha.in_off != 0 || Ha.out_level != 3 || Ha.out_off != 0 || Ha.act != XRL_ACT_NONE) return false;
    if (Hc.K != RH || Hc.N != 1 || Hc.in_level != 2 || Hc.in_off != RH || Hc.out_level != 3 || Hc.out_off != 2 || Hc.act != XRL_ACT_NONE) return false;
    return true;
}

int launch_rollout_fast(const xrl_rollout_step_t& p, int grid, hipStream_t stream) {
    XRL_ACT_DISPATCH(p.layers[0].act,
        if (p.n <= 256) hipLaunchKernelGGL((rollout_step_fast_kernel<ACT, 4>), dim3(grid), dim3(FUSED_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((rollout_step_fast_kernel<ACT, 16>), dim3(grid), dim3(FUSED_THREADS), 0, stream, p);)
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

}  // namespace xrl

namespace xrl { extern bool g_fast_enabled_ppo; }
extern "C" int xrl_set_fast_kernels(int enable) {
    xrl::g_fast_enabled = enable != 0;
    xrl::g_fast_enabled_ppo = enable != 0;
    return XRL_OK;
}
