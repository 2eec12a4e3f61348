// Fused PPO minibatch for the two-branch actor-critic D-256-256-{A | 1} with a Gaussian head (BASELINE configs[3], the
// MuJoCo network of configs/ppo/mujoco.yaml:8-13: Basic_Identical representation, actor_hidden_size = critic_hidden_size
// = [256, 256]).  ONE launch: rows -> both layers forward -> Gaussian PPO-clip loss -> backward incl. every weight
// gradient.  Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward),
// policies/gaussian.py + distributions.py:160-190 (DiagGaussianDistribution).
//
// Mapping (DESIGN.md section 3 has the measurements behind the rules):
//   * workgroup = (32-row tile, branch): the two branches share nothing but the rows, so a 4 096-row minibatch is
//     128 tiles x 2 = 256 workgroups, one per CU; both roles write disjoint parameter ranges of the tile's slab row.
//   * the 256 KB middle-layer weights arrive as a stream of B fragments (fragment-ordered copy kept current by the
//     optimiser launch, xrl_ppo_wide_pack) that is issued first and lands in registers: 32 float4 per lane.  A CU pulls
//     ~10 B/clk from L2, i.e. this stream lasts as long as the forward layer's matrix work.  Backward-data needs the
//     transposed fragments: a SECOND stream (the copy's backward section) into the same registers, issued the moment the
//     forward layer has consumed them -- it has the head / loss / weight-gradient phases (~30 k cycles, no global loads)
//     to arrive.  (First version: forward fragments transposed through LDS in eight 32 KB stages: 28-37 k cycles for
//     16 k cycles of matrix work, the stage writes and barriers could not be hidden.)
//   * 8 waves, wave w owns output columns [32 w, 32 w + 32) of every 256-wide product -- forward, backward-data and the
//     rows [32 w, ..) of dW1 -- so no split-K partial tiles are needed anywhere.
//   * first layer (K = D <= 24) on the matrix cores too; its weights are fetched as whole float4s of each wave's 32 rows and
//     handed over through LDS (per-lane fragment loads would be uncoalesced 4-byte loads in front of the weight stream).
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"
#include "rng.h"
#include "poststep.h"

#pragma clang fp contract(off)

namespace xrl {

constexpr int WH = 256;                       // hidden width
constexpr int WLD = WH + 4;                   // row stride of the 256-wide levels
constexpr int WQ = WH / 8;                    // k-chunks of the middle layer
constexpr int WQ_EARLY = 24;                  // backward chunks requested right after the forward layer
constexpr int WXLD = 28;                      // row stride of the observation tile (D <= 24, zero padded)
constexpr int WAM = 8;                        // max action dims
constexpr int W0S = 32 * 24;                  // first-layer rows of one wave's 32 output columns, staged in LDS (D <= 24)
constexpr int W_H1 = 0, W_H2 = W_H1 + FT * WLD, W_G2 = W_H2 + FT * WLD, W_XS = W_G2 + FT * WLD,
              W_RSC = W_XS + FT * WXLD + 8, W_DZH = W_RSC + FT * 16, W_IMG = W_DZH + FT * 16,
              W_STAT = W_IMG + WAM * WLD + 16, W_W0 = W_STAT + 5 * FT * 2, W_LDS_FLOATS = W_W0 + 8 * W0S;
constexpr int W_LDS_BYTES = W_LDS_FLOATS * 4;
static_assert(W_LDS_BYTES <= 160 * 1024, "tile does not fit the LDS of a CU");
static_assert((W_STAT % 2) == 0, "row statistics are doubles");

template <int CTRL>
__device__ __forceinline__ float wdpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// all-reduce over a row of 16 lanes (xor butterfly 8, 4, 2, 1 as row-rotate DPP moves)
__device__ __forceinline__ float wrow16_sum(float v) {
    v += wdpp<0x128>(v); v += wdpp<0x124>(v); v += wdpp<0x122>(v); v += wdpp<0x121>(v);
    return v;
}

// One B-fragment chunk through a buffer descriptor: the per-lane part of the address is ONE 32-bit register shared by all 32
// chunks, the chunk's place (wave-uniform: section, tile, rotated slot) rides in the scalar offset.  With flat loads hipcc
// keeps a 64-bit address pair per chunk alive (64 VGPRs next to the 128 the fragments occupy -> spills, and a spill of an
// in-flight fragment is a full s_waitcnt vmcnt(0) in the middle of the stream).
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wfrag_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const wu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wfrag_rsrc(const float* frag) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(frag), 0, 4 * WH * WH * 4, 0x00020000);
}
// byte offset of chunk q of output tile `wave` in section sec (0 forward, 1 backward) of branch `role`
__device__ __forceinline__ int wfrag_soff(int role, int sec, int wave, int q) {
    return ((role * 2 + sec) * WH * WH + (wave * WQ + frag_slot(q, wave, WQ, 1)) * 256) * 4;
}

#define WSTAMP(k) do { if (dbg_me) p.dbg[k] = clock64(); } while (0)

template <int ACT, int OACT>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_wide_kernel(xrl_ppo_wide_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds + W_H1;                            // [32][260] first hidden level; later dLoss/d(its pre-activation)
    float* h2 = lds + W_H2;                            // [32][260] second hidden level
    float* g2 = lds + W_G2;                            // [32][260] dLoss/d(pre-activation of h2)
    float* xs = lds + W_XS;                            // [32][28] observations (columns D.. zero)
    float* rsc = lds + W_RSC;                          // [32][16] action[0..7] | ret | adv | old_logp
    float* dzh = lds + W_DZH;                          // [32][16] dLoss/d(head pre-activations)[0..7] | d log_std terms[0..7]
    float* pimg = lds + W_IMG;                         // head weights [nout][260] | head bias[8] | log_std[8]
    double* rowstat = reinterpret_cast<double*>(lds + W_STAT);   // [5][32] per-row loss terms

    kernarg_prefetch<sizeof(xrl_ppo_wide_t)>();
    const int tid = threadIdx.x, M = p.M, D = p.D, A = p.A;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = blockIdx.x & 1, tile = blockIdx.x >> 1;        // role 0: actor, 1: critic
    const int m0 = tile * FT;
    const int r = tid >> 4, sub = tid & 15, m_row = m0 + r;
    const bool row_ok = m_row < M;
    xrl_wide_branch_t br;                               // (static indices: a run-time index would copy the struct to scratch)
    br.w0 = role ? p.br[1].w0 : p.br[0].w0; br.b0 = role ? p.br[1].b0 : p.br[0].b0;
    br.w1 = role ? p.br[1].w1 : p.br[0].w1; br.b1 = role ? p.br[1].b1 : p.br[0].b1;
    br.w2 = role ? p.br[1].w2 : p.br[0].w2; br.b2 = role ? p.br[1].b2 : p.br[0].b2;
    const int nout = role == 0 ? A : 1;
    float* slab = p.slabs + (size_t)tile * p.slab_stride;
    const bool dbg_me = p.dbg && tid == 0 && blockIdx.x == gridDim.x - 2 + (p.dbg_role & 1);
    WSTAMP(0);

    // ================= loads: everything small first (vmcnt retires in order), then the weight stream
    const int rows_here = min(FT, M - m0);
    float4 recv = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        // waves 0-2: observations as 8 D float4 of the contiguous [32][D] block; wave 3: actions; wave 4: the three scalars
        const float* src = nullptr; int c = 0, n_el = 0;
        if (tid < 192) { src = p.obs + (size_t)m0 * D; c = tid; n_el = rows_here * D; if (c >= 8 * D) src = nullptr; }
        else if (tid < 256) { src = p.actions + (size_t)m0 * A; c = tid - 192; n_el = rows_here * A; if (c >= 8 * A) src = nullptr; }
        if (src) {
            if (4 * c + 3 < n_el) recv = *reinterpret_cast<const float4*>(src + 4 * c);
            else {
                if (4 * c + 0 < n_el) recv.x = src[4 * c + 0];
                if (4 * c + 1 < n_el) recv.y = src[4 * c + 1];
                if (4 * c + 2 < n_el) recv.z = src[4 * c + 2];
            }
        }
        if (tid >= 256 && tid < 256 + FT && m0 + (tid - 256) < M) {
            const int m = m0 + tid - 256;
            recv.x = p.ret[m]; recv.y = p.adv[m]; recv.z = p.old_logp[m];
        }
    }
    float st_mean = 0.f, st_std = 1.f;
    if (p.stats) { st_mean = p.stats[0]; st_std = p.stats[1]; }
    float4 w2v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < nout * 64) w2v = *reinterpret_cast<const float4*>(p.params + br.w2 + (size_t)(tid >> 6) * WH + 4 * (tid & 63));
    float smallv = 0.f;
    if (tid >= 448 && tid < 448 + nout) smallv = p.params[br.b2 + tid - 448];
    if (tid >= 456 && tid < 456 + A) smallv = p.params[p.log_std_off + tid - 456];
    // first-layer weights of this wave's 32 output columns: one contiguous block of 32 D floats, fetched as whole float4s and
    // handed over through LDS (per-lane fragment loads would be 12 uncoalesced 4-byte loads in front of the weight stream)
    float4 w0v[3];
    {
        const float4* w0 = reinterpret_cast<const float4*>(p.params + br.w0 + (size_t)(wave * 32) * D);
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int c = lane + 64 * i; w0v[i] = c < 8 * D ? w0[c] : make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    const float b0v = p.params[br.b0 + wave * 32 + li], b1v = p.params[br.b1 + wave * 32 + li];
    float4 pf[WQ];                                      // B fragments of W1: output tile `wave`, all 32 k-chunks
    const __amdgpu_buffer_rsrc_t frs = wfrag_rsrc(p.frag);
#pragma unroll
    for (int q = 0; q < WQ; ++q) pf[q] = wfrag_load(frs, lane * 16, wfrag_soff(role, 0, wave, q));
    // hand the small things over through LDS
    if (tid < 192) {
        if (tid < 8 * D) {
            const float v4[4] = {recv.x, recv.y, recv.z, recv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = 4 * tid + i, rr = e / D, kk = e - rr * D; xs[rr * WXLD + kk] = v4[i]; }
        }
    } else if (tid < 256) {
        const int c = tid - 192;
        if (c < 8 * A) {
            const float v4[4] = {recv.x, recv.y, recv.z, recv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = 4 * c + i, rr = e / A, kk = e - rr * A; rsc[rr * 16 + kk] = v4[i]; }
        }
    } else if (tid < 256 + FT) {
        const int rr = tid - 256;
        rsc[rr * 16 + 8] = recv.x; rsc[rr * 16 + 9] = recv.y; rsc[rr * 16 + 10] = recv.z;
    } else if (tid >= 320) {                            // zero padding of the observation tile: columns D .. 27
        for (int e = tid - 320; e < FT * (WXLD - D); e += FUSED_THREADS - 320) {
            const int rr = e / (WXLD - D), kk = D + e - rr * (WXLD - D);
            xs[rr * WXLD + kk] = 0.f;
        }
    }
    if (tid < nout * 64) *reinterpret_cast<float4*>(pimg + (tid >> 6) * WLD + 4 * (tid & 63)) = w2v;
    if (tid >= 448 && tid < 448 + nout) pimg[WAM * WLD + tid - 448] = smallv;
    if (tid >= 456 && tid < 456 + A) pimg[WAM * WLD + 8 + tid - 456] = smallv;
    if (tid < 8) xs[FT * WXLD + tid] = 0.f;            // (tail the first-layer gradient's B operand may touch)
    {
        float* w0s = lds + W_W0 + wave * W0S;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int c = lane + 64 * i; if (c < 8 * D) *reinterpret_cast<float4*>(w0s + 4 * c) = w0v[i]; }
    }
    lds_barrier();                                                                                   // #0 rows, head image, W0
    WSTAMP(1);
    float4 w0f[3];                                      // B fragments of the first layer: W0[32 w + li][8 q + 4 lh + s], zero for k >= D
    {
        const float* w0r = lds + W_W0 + wave * W0S + li * D;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 8 * q + 4 * lh;
            w0f[q].x = k + 0 < D ? w0r[k + 0] : 0.f;
            w0f[q].y = k + 1 < D ? w0r[k + 1] : 0.f;
            w0f[q].z = k + 2 < D ? w0r[k + 2] : 0.f;
            w0f[q].w = k + 3 < D ? w0r[k + 3] : 0.f;
        }
    }

    // ================= forward: D -> 256 (three 8-wide k-chunks), 256 -> 256 (32 chunks from the register stream)
    {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* arow = xs + li * WXLD + 4 * lh;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 af = *reinterpret_cast<const float4*>(arow + q * 8);
            MFMA4(af, w0f[q], acc)
        }
        const int col = wave * 32 + li;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h1[row * WLD + col] = act_apply_c<ACT>(acc[rr] + b0v);
        }
    }
    lds_barrier();                                                                                   // #1 h1
    WSTAMP(2);
    {
        const float* arow = h1 + li * WLD + 4 * lh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int hq = 0; hq < WQ / 8; ++hq) {
            float4 af[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) af[q] = *reinterpret_cast<const float4*>(arow + (hq * 8 + q) * 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) { MFMA4(af[q], pf[hq * 8 + q], acc) }
        }
        const int col = wave * 32 + li;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h2[row * WLD + col] = act_apply_c<ACT>(acc[rr] + b1v);
        }
        // the forward fragments are consumed: the same registers take the backward section (W1 with the reduction index
        // n on the fragment's k axis: lane (li, lh) of chunk q holds W1[8 q + 4 lh + s][32 wave + li]), needed at dH1
        // (the scheduler must not lift these loads into the loop above: a fragment register would then have to hold its old
        //  and its new value at once, and the one that does not fit is spilled behind an s_waitcnt vmcnt(0))
        // Chunks 0 .. WQ_EARLY - 1 now; the rest when dH1 starts (their latency hides behind the first chunks' matrix work):
        // all 32 in flight through the loss / weight-gradient phases left those phases too few registers.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < WQ_EARLY; ++q) pf[q] = wfrag_load(frs, lane * 16, wfrag_soff(role, 1, wave, q));
        __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();                                                                                   // #2 h2
    WSTAMP(3);

    // ================= head forward (VALU, 16 threads per row), loss, head backward.  The 16 threads of a row all hold the
    // row's head pre-activations after the DPP all-reduce; the per-dimension work of the Gaussian loss (tanh, the log-prob
    // term, the gradient of mu and log_std) is spread over them -- thread `sub` owns action dim `sub` -- instead of
    // every thread evaluating all of it (measured: 15.2 k cycles for the actor role, 4 k for the critic's single output).
    {
        float4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4*>(h2 + r * WLD + 4 * (sub + 16 * i));
        float zmine = 0.f, z0 = 0.f;
#pragma unroll
        for (int j = 0; j < WAM; ++j) {
            if (j < nout) {
                float c = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 w = *reinterpret_cast<const float4*>(pimg + j * WLD + 4 * (sub + 16 * i));
                    c += a[i].x * w.x + a[i].y * w.y + a[i].z * w.z + a[i].w * w.w;
                }
                const float zj = wrow16_sum(c) + pimg[WAM * WLD + j];
                if (sub == j) zmine = zj;
                if (j == 0) z0 = zj;
            }
        }
        float my_dz = 0.f, my_gls = 0.f;                // dLoss/d(head pre-activation `sub`), d log_std term `sub` of this row
        double t_s = 0.0, t_c = 0.0, t_e = 0.0, t_v = 0.0, t_n = 0.0;
        const float invM = 1.f / (float)M;
        if (role == 0) {
            const bool mine = sub < A;
            float mu = 0.f, df = 0.f, var = 1.f, term = 0.f, entj = 0.f;
            if (mine) {
                mu = act_apply_c<OACT>(zmine);                                                      // activation_action
                const float ls = pimg[WAM * WLD + 8 + sub], sd = expf(ls);
                var = sd * sd; df = rsc[r * 16 + sub] - mu;
                term = -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;                          // Normal.log_prob of this dim
                entj = 0.5f + LOG_SQRT_2PI + logf(sd);                                              // Normal.entropy of this dim
            }
            const float logp = wrow16_sum(term), ent = wrow16_sum(entj);                            // summed over the action dims
            if (row_ok) {
                float adv = rsc[r * 16 + 9];
                const float old_lp = rsc[r * 16 + 10];
                asm volatile("" : "+v"(st_std));        // keeps hipcc from consuming the statistics (and waiting) at the top
                if (p.stats) adv = __fdiv_rn(__fsub_rn(adv, st_mean), st_std + 1e-8f);             // memory_tools.py:281-282
                const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
                const Surrogate s = surrogate(logp, old_lp, adv, lo, hi, invM);
                if (mine) {
                    const float gmu = s.dlogp * df / var;
                    my_gls = s.dlogp * (df * df / var - 1.f);
                    my_dz = gmu * act_grad_c<OACT>(mu);
                    if (p.heads) p.heads[(size_t)m_row * (A + 1) + sub] = mu;
                }
                t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                if (p.diag && sub == 0) {
                    const int m = m_row;
                    p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                }
            }
        } else if (row_ok) {
            const float v = z0, dv = v - rsc[r * 16 + 8];
            if (sub == 0) {
                my_dz = p.vf_coef * 2.f * dv * invM;                                                // d(vf * mean((v-ret)^2))/dv
                if (p.heads) p.heads[(size_t)m_row * (A + 1) + A] = v;
            }
            t_c = (double)dv * dv; t_v = v;
        }
        if (sub < 8) { dzh[r * 16 + sub] = my_dz; dzh[r * 16 + 8 + sub] = my_gls; }
        if (sub == 0) {
            rowstat[0 * FT + r] = t_s; rowstat[1 * FT + r] = t_c; rowstat[2 * FT + r] = t_e; rowstat[3 * FT + r] = t_v; rowstat[4 * FT + r] = t_n;
        }
        // the row's 16 threads sit in one wave and a wave's LDS operations execute in order: the values are there
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float4 d0 = *reinterpret_cast<const float4*>(dzh + r * 16), d1 = *reinterpret_cast<const float4*>(dzh + r * 16 + 4);
        const float dz[WAM] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        // dH2 = dZ . W2, times act'(h2): this thread's four k-chunks
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < WAM; ++j) {
                if (j < nout) {
                    const float4 w = *reinterpret_cast<const float4*>(pimg + j * WLD + 4 * (sub + 16 * i));
                    g.x += dz[j] * w.x; g.y += dz[j] * w.y; g.z += dz[j] * w.z; g.w += dz[j] * w.w;
                }
            }
            g.x *= act_grad_c<ACT>(a[i].x); g.y *= act_grad_c<ACT>(a[i].y); g.z *= act_grad_c<ACT>(a[i].z); g.w *= act_grad_c<ACT>(a[i].w);
            *reinterpret_cast<float4*>(g2 + r * WLD + 4 * (sub + 16 * i)) = g;
        }
    }
    lds_barrier();                                                                                   // #3 g2, dzh, rowstat
    WSTAMP(4);

    // ================= backward
    if (wave == 7) {                                    // loss terms of the tile (rows on lanes 0..31 of one wave)
        double acc_s = 0.0, acc_c = 0.0, acc_e = 0.0, acc_v = 0.0, acc_n = 0.0;
        if (lane < FT) { acc_s = rowstat[lane]; acc_c = rowstat[FT + lane]; acc_e = rowstat[2 * FT + lane]; acc_v = rowstat[3 * FT + lane]; acc_n = rowstat[4 * FT + lane]; }
        acc_s = wave_sum(acc_s); acc_c = wave_sum(acc_c); acc_e = wave_sum(acc_e); acc_v = wave_sum(acc_v); acc_n = wave_sum(acc_n);
        if (lane == 0) {
            double* q = p.partials + (size_t)blockIdx.x * 8;
            q[0] = acc_s; q[1] = acc_c; q[2] = acc_e; q[3] = acc_v; q[4] = acc_n; q[5] = 0; q[6] = 0; q[7] = 0;
        }
    }
    // ---- head weight gradients, head bias / log_std gradients, second-layer bias gradient: reductions over the 32 rows
    for (int o = tid; o < nout * WH; o += FUSED_THREADS) {
        const int j = o >> 8, k = o & (WH - 1);
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc += dzh[rr * 16 + j] * h2[rr * WLD + k];
        slab[br.w2 + o] = acc;
    }
    if (tid >= 256) {
        const int t = tid - 256;
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < FT; ++rr) acc += g2[rr * WLD + t];
        slab[br.b1 + t] = acc;
    } else if (tid < 16) {
        const int j = tid & 7, ls = tid >> 3;          // 0..7: head bias, 8..15: log_std (actor only)
        if (j < nout && (ls == 0 || role == 0)) {
            float acc = 0.f;
#pragma unroll
            for (int rr = 0; rr < FT; ++rr) acc += dzh[rr * 16 + tid];
            // d(-ent_coef * mean_m sum_j(log_std_j + c))/d log_std_j = -ent_coef, added once (tile 0)
            if (ls && tile == 0) acc -= p.ent_coef;
            slab[(ls ? p.log_std_off : br.b2) + j] = acc;
        }
    }
    WSTAMP(5);
    if (p.rows_g2) {
        // ---- the weight gradient of the middle layer is xrl_wide_dw1's: this tile's rows of g2 and h1 go to global memory (64 KB
        //      instead of a 256 KB partial of dW1; round 4 -- a CU moved ~800 KB per tile at its ~10 B/clk, 285 KB of them these
        //      stores, and the optimiser launch re-read all of it)
        const size_t base = ((size_t)role * (size_t)p.rows_ld + (size_t)m0) * WH;
        float4* G = reinterpret_cast<float4*>(p.rows_g2 + base);
        float4* Hr = reinterpret_cast<float4*>(p.rows_h1 + base);
#pragma unroll
        for (int i = 0; i < FT * (WH / 4) / FUSED_THREADS; ++i) {
            const int e = tid + i * FUSED_THREADS, rr = e >> 6, c4 = e & 63;
            G[e] = *reinterpret_cast<const float4*>(g2 + rr * WLD + 4 * c4);
            Hr[e] = *reinterpret_cast<const float4*>(h1 + rr * WLD + 4 * c4);
        }
    } else
    // ---- dW1[n][k] = sum_rows g2[row][n] * h1[row][k]: wave w owns rows n in [32 w, 32 w + 32), 8 column tiles in four
    //      passes of two (32 accumulator registers at a time: the backward fragments in flight hold 128)
    {
        const float* arow = g2 + lh * WLD + wave * 32 + li;             // A[i = n][k = row]
        const float* brow = h1 + lh * WLD + li;                         // B[k = row][j]
        float* dW = slab + br.w1 + (size_t)(wave * 32) * WH;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
            for (int s = 0; s < FT / 2; ++s) {
                const float av = arow[2 * s * WLD];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float bv = brow[2 * s * WLD + (pass * 2 + t) * 32];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                    dW[(size_t)row * WH + (pass * 2 + t) * 32 + li] = acc[t][rr];
                }
        }
    }
    WSTAMP(6);
    // ---- dH1 = g2 . W1, wave w owns output columns k in [32 w, 32 w + 32): the same loop as the forward layer with the
    //      backward fragments (in flight since the forward layer finished)
    {
        const float* arow = g2 + li * WLD + 4 * lh;
        const int k_out = wave * 32 + li;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = WQ_EARLY; q < WQ; ++q) pf[q] = wfrag_load(frs, lane * 16, wfrag_soff(role, 1, wave, q));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hq = 0; hq < WQ / 8; ++hq) {
            float4 af[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) af[q] = *reinterpret_cast<const float4*>(arow + (hq * 8 + q) * 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) { MFMA4(af[q], pf[hq * 8 + q], acc) }
        }
        lds_barrier();                                                  // every wave is done with h1 (dW1's B operand)
        // g1 = dH1 * act'(h1), in place
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            float* q = h1 + row * WLD + k_out;
            *q = acc[rr] * act_grad_c<ACT>(*q);
        }
    }
    lds_barrier();                                                                                   // #4 g1
    WSTAMP(7);
    // ---- first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k] on the matrix cores (wave w: rows c in [32 w, ..), one
    //      32-column tile of which the first D are kept), db0[c]
    {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* arow = h1 + lh * WLD + wave * 32 + li;             // A[i = c][k = row]
        const float* brow = xs + lh * WXLD + li;                        // B[k = row][j]: columns >= 28 alias the next row (discarded)
#pragma unroll
        for (int s = 0; s < FT / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * s * WLD], brow[2 * s * WXLD], acc, 0, 0, 0);
        if (li < D) {
            float* dW = slab + br.w0 + (size_t)(wave * 32) * D + li;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                dW[(size_t)row * D] = acc[rr];
            }
        }
        if (tid < WH) {
            float accb = 0.f;
#pragma unroll
            for (int rr = 0; rr < FT; ++rr) accb += h1[rr * WLD + tid];
            slab[br.b0 + tid] = accb;
        }
    }
    WSTAMP(8);
}

// ------------------------------------------------------------------------------------------------ acting step
// One launch for what the layered rollout step spends five on (xrl_obs_normalize, three grouped GEMM launches,
// xrl_policy_sample; reference: OnPolicyAgent.action / ppo_agent.py:97-135 -- policy(obs) -> dist.stochastic_sample(),
// log_prob, values; values of the next observations for the bootstrap).  A vector step has few rows (n = 128: 4 actor
// + 8 critic tiles) and a CU pulls only ~25 GB/s out of L2, so a workgroup that streamed a branch's whole 256 KB middle
// layer spent 10 us doing so (measured: 20 us per launch).  Each (tile, branch) is therefore given to WA_PARTS
// workgroups, one per 64 output columns of the middle layer (64 KB of fragments each); everything up to the head is
// column-local, the head's dot products are summed over the parts by whichever workgroup of the four finishes last
// (sc1 stores -> ticket counter -> sc1 loads, summed in part order: no spinning, no dependence on the arrival order).
// That workgroup samples and writes action / log-prob (actor) or the value (critic) -- the Philox draws and the
// arithmetic of policy_sample_kernel (rollout.hip), the log-prob summed over the action dims in the same order.
constexpr int WA_PARTS = 4;
constexpr int WA_H2LD = 68;                   // row stride of the 64-column slice of the second hidden level
constexpr int WA_RED = 8 * 32 * 33;           // K-quarter partial tiles of the slice: [4 quarters][2 column tiles][32][33]
constexpr int WA_OFF_RED = FT * WLD, WA_OFF_H2 = WA_OFF_RED + WA_RED, WA_OFF_XS = WA_OFF_H2 + FT * WA_H2LD,
              WA_OFF_IMG = WA_OFF_XS + FT * WXLD + 8, WA_OFF_TERMS = WA_OFF_IMG + WAM * WA_H2LD + 16,
              WA_OFF_W0 = WA_OFF_TERMS + FT * 8 + 4, WA_LDS_FLOATS = WA_OFF_W0 + 8 * W0S;
constexpr int WA_LDS_BYTES = WA_LDS_FLOATS * 4;
static_assert(WA_RED >= 2048 + 64 + 64 + 64, "statistics scratch lives in the partial-tile region");

template <int ACT, int OACT>
__global__ void __launch_bounds__(FUSED_THREADS) wide_act_kernel(xrl_wide_act_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h1 = lds;                                   // [32][260]
    float* red = lds + WA_OFF_RED;                     // partial tiles  (before the first layer: the statistics' scratch)
    float* h2p = lds + WA_OFF_H2;                      // [32][68] this part's 64 columns of the second hidden level
    float* xs = lds + WA_OFF_XS;                       // [32][28]
    float* pimg = lds + WA_OFF_IMG;                    // head weights of the slice [nout][68] | head bias[8] | log_std[8]
    float* terms = lds + WA_OFF_TERMS;                 // [32][8] per-dim log-prob terms
    int* s_last = reinterpret_cast<int*>(terms + FT * 8);

    kernarg_prefetch<sizeof(xrl_wide_act_t)>();
    if (p.has_post && blockIdx.x == gridDim.x - 1) {    // the PREVIOUS vector step's bookkeeping, beside the acting workgroups
        poststep_body<FUSED_THREADS>(p.post, reinterpret_cast<unsigned long long*>(lds));
        return;
    }
    const int tid = threadIdx.x, D = p.D, A = p.A, n = p.n;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool acting = (p.flags & 1) != 0, boot = (p.flags & 2) != 0;
    const int pair = (int)blockIdx.x / WA_PARTS, part = (int)blockIdx.x % WA_PARTS;
    const int tiles_a = acting ? (n + FT - 1) / FT : 0;
    const int role = pair < tiles_a ? 0 : 1;
    const int c_lo = acting ? 0 : n, c_hi = boot ? 2 * n : n;
    const int row0 = role == 0 ? pair * FT : c_lo + (pair - tiles_a) * FT;
    const int rows_here = min(FT, (role == 0 ? n : c_hi) - row0);
    const int r = tid >> 4, sub = tid & 15;
    xrl_wide_branch_t br;
    br.w0 = role ? p.br[1].w0 : p.br[0].w0; br.b0 = role ? p.br[1].b0 : p.br[0].b0;
    br.w1 = role ? p.br[1].w1 : p.br[0].w1; br.b1 = role ? p.br[1].b1 : p.br[0].b1;
    br.w2 = role ? p.br[1].w2 : p.br[0].w2; br.b2 = role ? p.br[1].b2 : p.br[0].b2;
    const int nout = role == 0 ? A : 1;
    const int tw = wave & 1, kq = wave >> 1;           // this wave's column tile of the slice and K-quarter in the middle layer
    // rows [0, n) may come as RAW observations: RunningMeanStd.update + _process_observation happen here then (every
    // workgroup that owns such rows forms the same statistics from all n rows; workgroup 0 stores them)
    const bool from_raw = p.raw != nullptr && row0 < n;
    const bool stats_writer = from_raw && blockIdx.x == 0;
    const bool dbg_me = p.dbg && tid == 0 && pair == 0;
#define ASTAMP(k) do { if (dbg_me) p.dbg[part * 8 + (k)] = clock64(); } while (0)
    ASTAMP(0);
    // rows [n, 2n) may come as the RAW next observations of the previous step: normalised here with the statistics as they are
    // (what xrl_rollout_poststep writes to x otherwise; get_terminated_values, on_policy.py:109)
    const bool from_next = p.next_raw != nullptr && row0 >= n;

    // ---- loads: small things first, then the weight stream
    constexpr int RV = 4;                               // rows per virtual thread of the statistics (host checks n <= RV * (1024 / D))
    const int R = 1024 / D;
    // Every load below is UNCONDITIONAL -- an index that is not wanted reads element 0 of a valid array and the value is dropped --
    // so that all of them are in flight together: written as `if (cond) v = p[i]` the compiler waits for each one before it
    // requests the next ("Loads the compiler can count"): eight dependent round trips at the top of a 13 us
    // launch that runs 256 times per rollout.
    float rawv[2][RV];
    float old_mean = 0.f, old_var = 1.f;
    double old_cnt = 0.0;
    {
        const bool want_raw = from_raw && p.update;
        const float* rawp = want_raw ? p.raw : p.params;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int vt = tid + v * FUSED_THREADS, d = vt % D, r0 = vt / D;
#pragma unroll
            for (int k = 0; k < RV; ++k) {
                const int rr = r0 + k * R;
                const bool ok = want_raw && r0 < R && rr < n;
                const float val = rawp[ok ? (size_t)rr * D + d : 0];
                rawv[v][k] = ok ? val : 0.f;
            }
        }
        const bool want_ms = (from_raw || from_next) && tid < D;
        const float* mp = want_ms ? p.mean_in : p.params;
        const float* vp = want_ms ? p.var_in : p.params;
        const float mv = mp[want_ms ? tid : 0], vv = vp[want_ms ? tid : 0];
        old_mean = want_ms ? mv : 0.f; old_var = want_ms ? vv : 1.f;
        const double* cp = from_raw ? p.count_in : reinterpret_cast<const double*>(p.params);
        const double cv = *cp;
        old_cnt = from_raw ? cv : 0.0;
    }
    float xv[2] = {0.f, 0.f};
    {
        const float* src = from_raw ? p.raw + (size_t)row0 * D : (from_next ? p.next_raw + (size_t)(row0 - n) * D : p.x + (size_t)row0 * D);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * FUSED_THREADS;
            const bool ok = e < rows_here * D;
            const float val = src[ok ? e : 0];
            xv[i] = ok ? val : 0.f;
        }
    }
    uint32_t step = p.step;
    {
        const uint32_t* sd = p.step_dev ? p.step_dev : reinterpret_cast<const uint32_t*>(p.params);
        const uint32_t sv = *sd;
        step += p.step_dev ? sv : 0u;
    }
    float4 w2v;
    {
        const bool ok = tid < nout * 16;
        const float4 val = *reinterpret_cast<const float4*>(p.params + br.w2 + (size_t)(ok ? (tid >> 4) : 0) * WH + 64 * part + 4 * (tid & 15));
        w2v = ok ? val : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float smallv;
    {
        const bool is_b2 = tid >= 448 && tid < 448 + nout, is_ls = tid >= 456 && tid < 456 + A;
        const float val = p.params[is_ls ? p.log_std_off + tid - 456 : (is_b2 ? br.b2 + tid - 448 : 0)];
        smallv = (is_b2 || is_ls) ? val : 0.f;
    }
    float4 w0v[3];
    {
        const float4* w0 = reinterpret_cast<const float4*>(p.params + br.w0 + (size_t)(wave * 32) * D);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            const float4 val = w0[c < 8 * D ? c : 0];
            w0v[i] = c < 8 * D ? val : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float b0v = p.params[br.b0 + wave * 32 + li];
    const float b1c = p.params[br.b1 + 64 * part + (tid & 63)];        // bias of the slice column this thread finishes below
    float4 pf[8];                                       // B fragments: column tile 2 part + tw of the middle layer, chunks 8 kq ..
    const __amdgpu_buffer_rsrc_t frs = wfrag_rsrc(p.frag);
#pragma unroll
    for (int q = 0; q < 8; ++q) pf[q] = wfrag_load(frs, lane * 16, wfrag_soff(role, 0, 2 * part + tw, 8 * kq + q));

    if (from_raw) {
        // RunningMeanStd.update (statistic_tools.py:117-185) with the arithmetic of rms_normalize_kernel (rollout.hip): its
        // 1 024 threads are this workgroup's threads twice over; every sum runs in the same order, in float64
        double* part_s = reinterpret_cast<double*>(red);               // [1024]
        double* bmean = part_s + 1024;                                 // [32]
        double* bvar = bmean + 32;                                     // [32]
        float* s_mean = reinterpret_cast<float*>(bvar + 32);           // [32]
        float* s_std = s_mean + 32;                                    // [32]
        if (p.update) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int vt = tid + v * FUSED_THREADS, r0 = vt / D;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < RV; ++k) if (r0 < R && r0 + k * R < n) acc += (double)rawv[v][k];
                part_s[vt] = r0 < R ? acc : 0.0;
            }
            lds_barrier();
            if (tid < D) {
                double t = 0.0;
                for (int rr = 0; rr < R; ++rr) t += part_s[rr * D + tid];
                bmean[tid] = (double)(float)(t / n);                   // np.mean returns float32
            }
            lds_barrier();
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int vt = tid + v * FUSED_THREADS, d = vt % D, r0 = vt / D;
                double q = 0.0;
                if (r0 < R) {
                    const double m = bmean[d];
#pragma unroll
                    for (int k = 0; k < RV; ++k) if (r0 + k * R < n) { const double df = (double)rawv[v][k] - m; q += df * df; }
                }
                part_s[vt] = r0 < R ? q : 0.0;
            }
            lds_barrier();
            if (tid < D) {
                double t = 0.0;
                for (int rr = 0; rr < R; ++rr) t += part_s[rr * D + tid];
                const float bstd = (float)sqrt(t / n);                 // np.std -> float32
                const float bv = bstd * bstd;                          // batch_var = np.square(batch_std)
                // update_from_moments (statistic_tools.py:173-185), float32 arrays with a Python-float count
                const double tot = old_cnt + (double)n;
                const float bm = (float)bmean[tid];
                const float delta = bm - old_mean;
                const float new_mean = old_mean + delta * (float)n / (float)tot;
                const float m_a = old_var * (float)old_cnt;
                const float m_b = bv * (float)n;
                const float M2 = m_a + m_b + (delta * delta) * (float)old_cnt * (float)n / (float)tot;
                const float new_var = M2 / (float)tot;
                if (stats_writer) { p.mean_out[tid] = new_mean; p.var_out[tid] = new_var; }
                s_mean[tid] = new_mean; s_std[tid] = sqrtf(new_var);
            }
            if (stats_writer && tid == 0) *p.count_out = old_cnt + (double)n;
        } else {
            if (tid < D) {
                s_mean[tid] = old_mean; s_std[tid] = sqrtf(old_var);
                if (stats_writer) { p.mean_out[tid] = old_mean; p.var_out[tid] = old_var; }
            }
            if (stats_writer && tid == 0) *p.count_out = old_cnt;
        }
    } else if (from_next) {
        float* s_mean = red + 2 * (1024 + 64);
        if (tid < D) { s_mean[tid] = old_mean; s_mean[32 + tid] = sqrtf(old_var); }
    }
    if (from_raw || from_next) {
        const float* s_mean = red + 2 * (1024 + 64);
        const float* s_std = s_mean + 32;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * FUSED_THREADS;
            if (e < rows_here * D) {
                const int d = e % D;
                float v = xv[i];
                if (p.normalize) {
                    v = (v - s_mean[d]) / (s_std[d] + 1e-8f);          // _process_observation (agent.py:262-283)
                    v = fminf(fmaxf(v, -p.range), p.range);
                }
                xv[i] = v;
                if (from_raw && p.obs_slot && part == 0 && role == 0) p.obs_slot[(size_t)row0 * D + e] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * FUSED_THREADS;
        if (e < FT * D) { const int rr = e / D, kk = e - rr * D; xs[rr * WXLD + kk] = xv[i]; }
    }
    for (int e = tid; e < FT * (WXLD - D); e += FUSED_THREADS) {       // zero padding: columns D .. 27
        const int rr = e / (WXLD - D), kk = D + e - rr * (WXLD - D);
        xs[rr * WXLD + kk] = 0.f;
    }
    if (tid < nout * 16) *reinterpret_cast<float4*>(pimg + (tid >> 4) * WA_H2LD + 4 * (tid & 15)) = w2v;
    if (tid >= 448 && tid < 448 + nout) pimg[WAM * WA_H2LD + tid - 448] = smallv;
    if (tid >= 456 && tid < 456 + A) pimg[WAM * WA_H2LD + 8 + tid - 456] = smallv;
    {
        float* w0s = lds + WA_OFF_W0 + wave * W0S;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int c = lane + 64 * i; if (c < 8 * D) *reinterpret_cast<float4*>(w0s + 4 * c) = w0v[i]; }
    }
    lds_barrier();
    ASTAMP(1);
    float4 w0f[3];                                      // B fragments of the first layer: W0[32 w + li][8 q + 4 lh + s], zero for k >= D
    {
        const float* w0r = lds + WA_OFF_W0 + wave * W0S + li * D;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 8 * q + 4 * lh;
            w0f[q].x = k + 0 < D ? w0r[k + 0] : 0.f;
            w0f[q].y = k + 1 < D ? w0r[k + 1] : 0.f;
            w0f[q].z = k + 2 < D ? w0r[k + 2] : 0.f;
            w0f[q].w = k + 3 < D ? w0r[k + 3] : 0.f;
        }
    }

    // ---- first layer: all 256 columns (every part needs them), same product and epilogue as ppo_wide_kernel
    {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* arow = xs + li * WXLD + 4 * lh;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 af = *reinterpret_cast<const float4*>(arow + q * 8);
            MFMA4(af, w0f[q], acc)
        }
        const int col = wave * 32 + li;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            h1[row * WLD + col] = act_apply_c<ACT>(acc[rr] + b0v);
        }
    }
    lds_barrier();
    ASTAMP(2);
    // ---- middle layer, this part's 64 columns: wave = (column tile tw, K-quarter kq); the four quarters meet in LDS and are
    //      added in quarter order
    {
        const float* arow = h1 + li * WLD + 4 * lh + kq * 64;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float4 af[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) af[q] = *reinterpret_cast<const float4*>(arow + q * 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { MFMA4(af[q], pf[q], acc) }
        float* dst = red + (kq * 2 + tw) * (32 * 33);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            dst[row * 33 + li] = acc[rr];
        }
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * FUSED_THREADS, row = e >> 6, c = e & 63, t2 = c >> 5, cc = c & 31;
        float v = red[(0 * 2 + t2) * (32 * 33) + row * 33 + cc];
        v += red[(1 * 2 + t2) * (32 * 33) + row * 33 + cc];
        v += red[(2 * 2 + t2) * (32 * 33) + row * 33 + cc];
        v += red[(3 * 2 + t2) * (32 * 33) + row * 33 + cc];
        h2p[row * WA_H2LD + c] = act_apply_c<ACT>(v + b1c);           // (c == tid & 63 for every i)
    }
    lds_barrier();
    ASTAMP(3);
    // ---- head: partial dot products over this part's 64 columns (16 threads per row, 4 columns each), published for the
    //      last workgroup of the pair
    const int e = row0 + r;
    const bool row_ok = r < rows_here;
    {
        const float4 a = *reinterpret_cast<const float4*>(h2p + r * WA_H2LD + 4 * sub);
        float zp = 0.f;
#pragma unroll
        for (int j = 0; j < WAM; ++j) {
            if (j < nout) {
                const float4 w = *reinterpret_cast<const float4*>(pimg + j * WA_H2LD + 4 * sub);
                const float zj = wrow16_sum(a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w);
                if (sub == j) zp = zj;
            }
        }
        if (sub < 8) __hip_atomic_store(p.xchg + (((size_t)pair * WA_PARTS + part) * FT + r) * 8 + sub, zp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the partial sums are out (acknowledged) before the ticket is taken
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.xcnt + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = t == WA_PARTS - 1;
        if (t == WA_PARTS - 1) __hip_atomic_store(p.xcnt + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    }
    __syncthreads();
    ASTAMP(4);
    if (!*s_last) return;
    float zmine = 0.f;                                  // head pre-activation `sub` of row r (actor), value (critic: sub 0)
    if (sub < 8) {
        const float* src = p.xchg + ((size_t)pair * WA_PARTS * FT + r) * 8 + sub;
#pragma unroll
        for (int c = 0; c < WA_PARTS; ++c) zmine += __hip_atomic_load(src + (size_t)c * FT * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        zmine += pimg[WAM * WA_H2LD + sub];
    }
    if (role == 0) {
        if (row_ok && sub < A) {
            const int j = sub;
            const float mu = act_apply_c<OACT>(zmine);
            const float zn = policy_normal(p.seed, (uint32_t)e, step, (uint32_t)j);
            const float ls = pimg[WAM * WA_H2LD + 8 + j], sd = expf(ls);
            const float x = mu + sd * zn;                  // Normal(mu, std).sample()
            const float df = x - mu;
            terms[r * 8 + j] = -(df * df) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI;
            p.act_out[(size_t)e * A + j] = x;
            if (p.env_action_f) p.env_action_f[(size_t)e * A + j] = x;
        }
        lds_barrier();
        if (row_ok && sub == 0) {
            float logp = 0.f;
            for (int j = 0; j < A; ++j) logp += terms[r * 8 + j];
            p.logp_out[e] = logp;
        }
    } else if (row_ok && sub == 0) {
        if (e < n) { if (p.val_out) p.val_out[e] = zmine; }
        else if (p.bootv_prev) p.bootv_prev[e - n] = zmine;
    }
    ASTAMP(5);
#undef ASTAMP
}

// frag[b][0][tile t][slot (q + t) mod 32][lane l][s] = W1_b[32 t + (l & 31)][8 q + 4 (l >> 5) + s]      (forward section)
// frag[b][1][tile t][slot (q + t) mod 32][lane l][s] = W1_b[8 q + 4 (l >> 5) + s][32 t + (l & 31)]      (backward section)
__global__ void __launch_bounds__(256) ppo_wide_pack_kernel(xrl_ppo_wide_t p, float* __restrict__ frag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                // one float4 of the destination
    if (i >= 4 * WH * WH / 4) return;
    const int sec = i / (WH * WH / 4), e = i - sec * (WH * WH / 4);
    const int b = sec >> 1, bwd = sec & 1;
    const int l = e & 63, slot = (e >> 6) & (WQ - 1), t = e >> 11;
    const int q = (slot - t + WQ) & (WQ - 1);
    const int w1 = b ? p.br[1].w1 : p.br[0].w1;
    float4 v;
    if (!bwd) {
        v = *reinterpret_cast<const float4*>(p.params + w1 + (size_t)(32 * t + (l & 31)) * WH + 8 * q + 4 * (l >> 5));
    } else {
        const float* src = p.params + w1 + (size_t)(8 * q + 4 * (l >> 5)) * WH + 32 * t + (l & 31);
        v = make_float4(src[0], src[WH], src[2 * WH], src[3 * WH]);
    }
    *reinterpret_cast<float4*>(frag + (size_t)i * 4) = v;
}

static int wide_check(const xrl_ppo_wide_t* p) {
    XRL_CHECK_ARG(p != nullptr && p->params && p->H == WH && p->D >= 1 && p->D <= 24 && p->A >= 1 && p->A <= WAM);
    for (int b = 0; b < 2; ++b) XRL_CHECK_ARG(p->br[b].w0 % 4 == 0 && p->br[b].w1 % 4 == 0 && p->br[b].w2 % 4 == 0);
    return XRL_OK;
}

#define WIDE_FOR_EACH(X) X(XRL_ACT_RELU, XRL_ACT_NONE) X(XRL_ACT_RELU, XRL_ACT_TANH) X(XRL_ACT_LEAKY_RELU, XRL_ACT_NONE) \
    X(XRL_ACT_LEAKY_RELU, XRL_ACT_TANH) X(XRL_ACT_TANH, XRL_ACT_NONE) X(XRL_ACT_TANH, XRL_ACT_TANH)

int init_ppo_wide() {
#define WIDE_ATTR(a, o) XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_wide_kernel<a, o>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES));
    WIDE_FOR_EACH(WIDE_ATTR)
#undef WIDE_ATTR
#define WIDE_ATTR(a, o) XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_act_kernel<a, o>), hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS_BYTES));
    WIDE_FOR_EACH(WIDE_ATTR)
#undef WIDE_ATTR
    return XRL_OK;
}

static bool wide_act_ok(int act, int out_act) {
    return (act == XRL_ACT_RELU || act == XRL_ACT_LEAKY_RELU || act == XRL_ACT_TANH) && (out_act == XRL_ACT_NONE || out_act == XRL_ACT_TANH);
}

// dW1 of both branches from the rows ppo_wide_kernel left in global memory (xrl_wide_dw1): workgroup = (part of DW_ROWS rows, branch,
// 128 x 128 quarter of the 256 x 256 matrix); the part's rows of g2 (its 128 columns n) and h1 (its 128 columns k) staged in LDS,
// then the weight-gradient loop of the minibatch kernels: wave w owns the 32-row block (w & 3) of n and two 32-column blocks of k,
// DW_ROWS / 2 chained MFMAs per block, rows in order.  8 * parts workgroups (4 096 rows: 256), 128 KB read + 16.4 k cycles of
// matrix work + 64 KB written each.  (Measured, round 4: requesting the rows as two halves and starting the first half's MFMAs while
// the second is on its way does not shorten it -- 16.5 vs 15.6 us at 4 096 rows; the MFMA operands straight from global memory,
// without the staging: 25.4 us.)
constexpr int DW_ROWS = 128, DW_LD = 128 + 4;
constexpr int DW_LDS_BYTES = 2 * DW_ROWS * DW_LD * 4;
__global__ void __launch_bounds__(FUSED_THREADS) wide_dw1_kernel(xrl_ppo_wide_t p, int rows, int parts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sG = lds;
    float* sH = lds + DW_ROWS * DW_LD;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = blockIdx.x % parts, q = blockIdx.x / parts, kt = q & 1, nt = (q >> 1) & 1, role = q >> 2;
    const int r0 = part * DW_ROWS;
    const size_t base = ((size_t)role * (size_t)p.rows_ld + (size_t)r0) * WH;
    const float* G = p.rows_g2 + base + nt * 128;
    const float* Hh = p.rows_h1 + base + kt * 128;
#pragma unroll
    for (int i = 0; i < DW_ROWS * 32 / FUSED_THREADS; ++i) {
        const int e = tid + i * FUSED_THREADS, rr = e >> 5, c4 = e & 31;
        // (unconditional loads from a clamped row, zeroed by a select: a branch around each load makes hipcc wait for every one
        //  of them in turn -- eight dependent round trips)
        const bool ok = r0 + rr < rows;
        const size_t o = (size_t)(ok ? rr : 0) * WH + 4 * c4;
        float4 g = *reinterpret_cast<const float4*>(G + o), h = *reinterpret_cast<const float4*>(Hh + o);
        if (!ok) { g = make_float4(0.f, 0.f, 0.f, 0.f); h = g; }
        *reinterpret_cast<float4*>(sG + rr * DW_LD + 4 * c4) = g;
        *reinterpret_cast<float4*>(sH + rr * DW_LD + 4 * c4) = h;
    }
    __syncthreads();
    const int a = wave & 3, b0 = 2 * (wave >> 2);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const float* arow = sG + lh * DW_LD + a * 32 + li;                  // A[i = n][k = row]
    const float* brow = sH + lh * DW_LD + b0 * 32 + li;                 // B[k = row][j = k column]
#pragma unroll 8
    for (int s = 0; s < DW_ROWS / 2; ++s) {
        const float av = arow[2 * s * DW_LD];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[2 * s * DW_LD + t * 32], acc[t], 0, 0, 0);
    }
    const int w1 = role ? p.br[1].w1 : p.br[0].w1;
    float* dW = p.slabs + (size_t)part * p.slab_stride + w1 + (size_t)(nt * 128 + a * 32) * WH + kt * 128;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            dW[(size_t)row * WH + (b0 + t) * 32 + li] = acc[t][rr];
        }
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_ppo_wide_pack(const xrl_ppo_wide_t* p, float* frag, xrl_stream_t stream) {
    int rc = wide_check(p);
    if (rc != XRL_OK) return rc;
    XRL_CHECK_ARG(frag != nullptr && (reinterpret_cast<uintptr_t>(p->params) & 15) == 0);
    hipLaunchKernelGGL(ppo_wide_pack_kernel, dim3(4 * WH * WH / 4 / 256), dim3(256), 0, as_stream(stream), *p, frag);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_ppo_wide_minibatch(const xrl_ppo_wide_t* p, xrl_stream_t stream) {
    int rc = wide_check(p);
    if (rc != XRL_OK) return rc;
    XRL_CHECK_ARG(p->frag && p->obs && p->actions && p->ret && p->adv && p->old_logp && p->slabs && p->partials && p->M > 0);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p->params) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->frag) & 15) == 0);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p->obs) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->actions) & 15) == 0);
    static bool inited = false;
    if (!inited) {
        rc = init_ppo_wide();
        if (rc != XRL_OK) return rc;
        inited = true;
    }
    const int n_tiles = (p->M + FT - 1) / FT;
    XRL_CHECK_ARG(wide_act_ok(p->act, p->out_act));
#define WIDE_LAUNCH(a, o)                                                                                                     \
    if (p->act == a && p->out_act == o)                                                                                       \
        hipLaunchKernelGGL((ppo_wide_kernel<a, o>), dim3(2 * n_tiles), dim3(FUSED_THREADS), W_LDS_BYTES, as_stream(stream), *p);
    WIDE_FOR_EACH(WIDE_LAUNCH)
#undef WIDE_LAUNCH
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_wide_dw1(const xrl_ppo_wide_t* p, int32_t* n_parts, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->rows_g2 && p->rows_h1 && p->slabs && p->M > 0 && p->H == WH);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p->rows_g2) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->rows_h1) & 15) == 0);
    const int rows = ((p->M + FT - 1) / FT) * FT, parts = (rows + DW_ROWS - 1) / DW_ROWS;
    XRL_CHECK_ARG(p->rows_ld >= rows);
    static bool inited = false;
    if (!inited) {
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_dw1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES));
        inited = true;
    }
    hipLaunchKernelGGL(wide_dw1_kernel, dim3(8 * parts), dim3(FUSED_THREADS), DW_LDS_BYTES, as_stream(stream), *p, rows, parts);
    XRL_CHECK_LAUNCH();
    if (n_parts) *n_parts = parts;
    return XRL_OK;
}

extern "C" int xrl_wide_act_step(const xrl_wide_act_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p != nullptr && p->params && p->frag && p->x && p->H == WH && p->D >= 1 && p->D <= 24 && p->A >= 1 && p->A <= WAM);
    XRL_CHECK_ARG(p->n > 0 && (p->flags & 3) != 0 && wide_act_ok(p->act, p->out_act));
    XRL_CHECK_ARG(!(p->flags & 1) || (p->act_out && p->logp_out));
    if (p->raw) XRL_CHECK_ARG((p->flags & 1) && p->mean_in && p->var_in && p->count_in && p->mean_out && p->var_out && p->count_out &&
                              p->mean_in != p->mean_out && p->n <= 4 * (1024 / p->D) &&
                              (p->n % FT == 0 || !(p->flags & 2)));   // a critic tile must not hold raw and normalised rows
    XRL_CHECK_ARG(!(p->flags & 2) || p->bootv_prev);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p->params) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->frag) & 15) == 0);
    for (int b = 0; b < 2; ++b) XRL_CHECK_ARG(p->br[b].w0 % 4 == 0 && p->br[b].w1 % 4 == 0 && p->br[b].w2 % 4 == 0);
    static bool inited = false;
    if (!inited) {
        int rc = init_ppo_wide();
        if (rc != XRL_OK) return rc;
        inited = true;
    }
    const bool acting = p->flags & 1, boot = p->flags & 2;
    const int tiles_a = acting ? (p->n + FT - 1) / FT : 0;
    const int c_rows = (boot ? 2 * p->n : p->n) - (acting ? 0 : p->n);
    const int grid = (tiles_a + (c_rows + FT - 1) / FT) * WA_PARTS + (p->has_post ? 1 : 0);
    if (p->next_raw) XRL_CHECK_ARG(boot && p->mean_in && p->var_in && p->n % FT == 0);
    if (p->has_post) {
        const xrl_poststep_t& q = p->post;
        XRL_CHECK_ARG(q.reward && q.terminated && q.truncated && q.rew_out && q.term_out && q.seg_out && q.ret_track && q.ret_mean &&
                      q.ret_var && q.ret_count && q.n > 0 && q.D > 0 && (q.next_obs_norm == nullptr || (q.next_obs && (!q.use_obsnorm || (q.obs_mean && q.obs_var)))));
    }
    XRL_CHECK_ARG(p->xchg && p->xcnt);
#define WIDE_LAUNCH(a, o)                                                                                                     \
    if (p->act == a && p->out_act == o)                                                                                       \
        hipLaunchKernelGGL((wide_act_kernel<a, o>), dim3(grid), dim3(FUSED_THREADS), WA_LDS_BYTES, as_stream(stream), *p);
    WIDE_FOR_EACH(WIDE_LAUNCH)
#undef WIDE_LAUNCH
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
