// The tail of an on-policy acting step on a device environment (xrl_act_tail, include/xrl_hip.h) as ONE launch: the general path of
// PPO_Agent's vector step (any actor-critic MLP with a shared trunk, any device env -- configs/ppo/classic_control/*.yaml) was seven
// launches of ~6 us: normalise + store, three grouped products, xrl_policy_sample, the env's step, xrl_rollout_poststep.  The last
// product (the heads: A logits / means on the actor's features, one value on the critic's), the sampling and the env step need nothing of
// another env; here a workgroup of two waves takes 16 envs:
//   thread (row, branch), row = env e (this step's observation) or n + e (the previous step's next observation, whose value bootstraps
//   cut paths): the head outputs of that row and branch -- every output the fma chain xrl_linear_fwd's MFMA tile makes of it (K in slabs
//   of 32; inside a slab k = 8 q + s then 8 q + 4 + s for q, s = 0..3: the lane halves of v_mfma_f32_32x32x2_f32, gemm.hip), + bias --
//   from the rows' features staged in LDS (coalesced 16-byte loads, row stride 257: conflict-free), the head weights in LDS as well
//   (broadcast reads); wave 0 = the actor's outputs, wave 1 = the critic's;
//   thread e < 16: policy_sample_one (xrl_policy_sample's statements), then the env's step (classic_step_one / cartpole_step_one).
// With xrl_post_norm (the previous step's bookkeeping + this step's observation statistics as one launch) a vector step is four launches.
// Same numbers as the launches it replaces, bit for bit (tests/test_gpu_agent.py).  Reference: on_policy.py:128-169 (get_actions),
// ppo_agent.py:113-143, actor_head.py / critic_head.py (the heads), Gymnasium's classic_control dynamics (csrc/classic.h, cartpole.h).
#include "common.h"
#include "rng.h"
#include "cartpole.h"
#include "classic.h"
#include "sample.h"

namespace xrl {

constexpr int AT_THREADS = 128, AT_ENVS = 16, AT_MAXK = 128, AT_LD = 2 * AT_MAXK + 1, AT_MAXA = 8, AT_LDW = AT_MAXK + 1;

// NO outputs of one row: acc_j = the fma chain xrl_linear_fwd's 64 x 64 MFMA tile makes of output j (gemm.hip, MODE_NT: K in slabs of 32;
// inside a slab the instruction (q, s) multiplies k = 8 q + s in the lower lane half and k = 8 q + 4 + s in the upper one)
template <int NO>
__device__ __forceinline__ void at_heads_row(const float* x, const float* w, int K, float (&acc)[AT_MAXA]) {
    for (int k0 = 0; k0 < K; k0 += 32) {
        float xv[32], wv[NO][32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            xv[i] = x[k0 + i];
#pragma unroll
            for (int j = 0; j < NO; ++j) wv[j][i] = w[j * AT_LDW + k0 + i];            // (the same address in every lane: broadcast reads)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 8 * q + 4 * h + s;
#pragma unroll
                    for (int j = 0; j < NO; ++j) acc[j] = fmaf(xv[i], wv[j][i], acc[j]);
                }
            }
        }
    }
}

__device__ long long* g_at_dbg = nullptr;      // tools/probe_act_tail.py: clock stamps of workgroup 0 (xrl_debug_act_tail_stamps)
#define ATSTAMP(k) do { if (adbg && threadIdx.x == 0 && blockIdx.x == 0) adbg[k] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

__global__ void __launch_bounds__(AT_THREADS) act_tail_kernel(xrl_act_tail_t p) {
    long long* const adbg = g_at_dbg;
    ATSTAMP(0);
    __shared__ __attribute__((aligned(16))) float s_feat[2 * AT_ENVS * AT_LD];     // rows e, then rows n + e
    __shared__ float s_w[(AT_MAXA + 1) * AT_LDW];                                // the A actor rows, then the critic row
    __shared__ float s_heads[2 * AT_ENVS][AT_MAXA + 2];
    const int tid = threadIdx.x, n = p.sample.n, A = p.sample.A, K = p.K;
    const int e0 = blockIdx.x * AT_ENVS, ne = min(AT_ENVS, n - e0);
    const int W = p.a_off < p.c_off ? p.c_off + K : p.a_off + K;            // feature columns a row needs (both branches)
    // ---- the rows' features -> LDS: row r < ne is global row e0 + r, row AT_ENVS + r is global row n + e0 + r (boot rows: only if read)
    // (every load is issued before the first LDS store: as a plain loop -- trip count unknown to the compiler -- each of the 16 quads of a
    //  thread was a global round trip of its own, ~10 us for the launch)
    {
        const int q_per_row = (W + 3) >> 2, halves = p.boot_rows ? 2 : 1;
        const int total = halves * ne * q_per_row;
        constexpr int NQ = 2 * AT_ENVS * (2 * AT_MAXK / 4) / AT_THREADS;    // 16 quads per thread at most
        float4 v[NQ];
        int dsto[NQ];
        const bool vec = (p.ldh & 3) == 0 && (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.hb) & 15) == 0);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = tid + u * AT_THREADS;
            dsto[u] = -1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                const int r = i / q_per_row, q = i - r * q_per_row;
                const int half = r >= ne ? 1 : 0, rr = r - half * ne;
                const float* src = p.hb + (size_t)(half * n + e0 + rr) * p.ldh + 4 * q;
                dsto[u] = (half * AT_ENVS + rr) * AT_LD + 4 * q;
                if (vec) v[u] = *reinterpret_cast<const float4*>(src);
                else {
                    if (4 * q + 0 < W) v[u].x = src[0];
                    if (4 * q + 1 < W) v[u].y = src[1];
                    if (4 * q + 2 < W) v[u].z = src[2];
                    if (4 * q + 3 < W) v[u].w = src[3];
                }
            }
        }
        constexpr int NWQ = ((AT_MAXA + 1) * AT_MAXK + AT_THREADS - 1) / AT_THREADS;   // 9 weights per thread at most
        float wv[NWQ];
#pragma unroll
        for (int u = 0; u < NWQ; ++u) {
            const int i = tid + u * AT_THREADS;
            wv[u] = 0.f;
            if (i < (A + 1) * K) { const int j = i / K, k = i - j * K; wv[u] = j < A ? p.w_actor[(size_t)j * p.ldw_a + k] : p.w_critic[k]; }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if (dsto[u] >= 0) { float* d = s_feat + dsto[u]; d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
#pragma unroll
        for (int u = 0; u < NWQ; ++u) {
            const int i = tid + u * AT_THREADS;
            if (i < (A + 1) * K) { const int j = i / K, k = i - j * K; s_w[j * AT_LDW + k] = wv[u]; }
        }
    }
    __syncthreads();
    ATSTAMP(1);
    // ---- heads: thread (row, branch); wave 0: the actor's A outputs of the 32 rows, wave 1: the critic's value
    {
        const int row = tid & 63, branch = __builtin_amdgcn_readfirstlane(tid >> 6);   // rows [0, 2 * AT_ENVS = 32) live
        const int half = row >= AT_ENVS ? 1 : 0, rr = row - half * AT_ENVS;
        const bool live = row < 2 * AT_ENVS && rr < ne && (half == 0 || p.boot_rows) && (branch == 1 || half == 0 || p.boot_actor);
        if (live) {
            const float* x = s_feat + (size_t)row * AT_LD + (branch ? p.c_off : p.a_off);
            const float* w = s_w + (branch ? A * AT_LDW : 0);
            const int no = branch ? 1 : A;
            float acc[AT_MAXA];
#pragma unroll
            for (int j = 0; j < AT_MAXA; ++j) acc[j] = 0.f;
            switch (no) {                                                   // (the output count as a compile-time number: no branch per fma)
                case 1: at_heads_row<1>(x, w, K, acc); break;
                case 2: at_heads_row<2>(x, w, K, acc); break;
                case 3: at_heads_row<3>(x, w, K, acc); break;
                case 4: at_heads_row<4>(x, w, K, acc); break;
                case 5: at_heads_row<5>(x, w, K, acc); break;
                case 6: at_heads_row<6>(x, w, K, acc); break;
                case 7: at_heads_row<7>(x, w, K, acc); break;
                default: at_heads_row<8>(x, w, K, acc); break;
            }
            const float* b = branch ? p.b_critic : p.b_actor;
#pragma unroll
            for (int j = 0; j < AT_MAXA; ++j)
                if (j < no) {
                    float y = acc[j] + (b ? b[j] : 0.f);
                    if (!branch) { const float z = y; XRL_ACT_DISPATCH(p.act_actor, y = act_apply_c<ACT>(z);) }   // activation_action (actor_head.py:62)
                    s_heads[row][branch ? A : j] = y;
                    if (p.heads) p.heads[(size_t)(half * n + e0 + rr) * (A + 1) + (branch ? A : j)] = y;
                }
        }
    }
    __syncthreads();
    ATSTAMP(2);
    // ---- one thread per env: sample + store, then the env's step
    if (tid < ne) {
        const int e = e0 + tid;
        policy_sample_one(p.sample, e, s_heads[tid], p.boot_rows ? s_heads[AT_ENVS + tid][A] : 0.f);
        ATSTAMP(3);
        if (p.sample.act_out) {
            if (p.env_kind == 4) cartpole_step_one(p.cartpole, e);
            else if (p.env_kind >= 1 && p.env_kind <= 3) classic_step_one(p.classic, e);
        }
    }
    ATSTAMP(4);
}
#undef ATSTAMP

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_debug_act_tail_stamps(long long* stamps) {
    XRL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_at_dbg), &stamps, sizeof(stamps)));
    return XRL_OK;
}

extern "C" int xrl_act_tail(const xrl_act_tail_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_act_tail_t& p = *pp;
    const xrl_sample_t& q = p.sample;
    XRL_CHECK_ARG(p.hb && p.w_actor && p.w_critic && q.n > 0 && q.A >= 1 && q.A <= AT_MAXA && q.ld == q.A + 1);
    XRL_CHECK_ARG(p.K >= 32 && p.K <= AT_MAXK && (p.K & 31) == 0 && p.a_off >= 0 && p.c_off >= 0 &&
                  (p.a_off + p.K <= p.c_off || p.c_off + p.K <= p.a_off) && p.a_off + p.K <= 2 * AT_MAXK && p.c_off + p.K <= 2 * AT_MAXK);
    XRL_CHECK_ARG(p.ldh >= (p.a_off < p.c_off ? p.c_off : p.a_off) + p.K && p.ldw_a >= p.K && p.ldw_c >= p.K);
    XRL_CHECK_ARG((q.act_out && q.logp_out) || (!q.act_out && q.bootv_prev));
    XRL_CHECK_ARG(!q.bootv_prev || p.boot_rows);
    XRL_CHECK_ARG(!q.gaussian || q.log_std);
    XRL_CHECK_ARG(p.env_kind >= 0 && p.env_kind <= 4 && p.act_actor >= 0);
    if (q.act_out && p.env_kind == 4) {
        const xrl_cartpole_t& c = p.cartpole;
        XRL_CHECK_ARG(c.state && c.steps && c.episodes && c.obs && c.ep_score && c.n == q.n && c.action && c.next_obs && c.reward &&
                      c.terminated && c.truncated && c.stats);
    } else if (q.act_out && p.env_kind >= 1) {
        const xrl_classic_t& c = p.classic;
        XRL_CHECK_ARG(c.kind == p.env_kind && c.n == q.n && c.max_steps > 0 && c.state && c.steps && c.episodes && c.obs && c.ep_score &&
                      c.next_obs && c.reward && c.terminated && c.truncated && c.stats);
        XRL_CHECK_ARG(c.kind == CLASSIC_PENDULUM ? c.action_f != nullptr : c.action != nullptr);
    }
    hipLaunchKernelGGL(act_tail_kernel, dim3((q.n + AT_ENVS - 1) / AT_ENVS), dim3(AT_THREADS), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
