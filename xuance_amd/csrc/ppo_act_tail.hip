// The tail of an on-policy acting pass on frame stacks (xrl_ppo_act_tail, include/xrl_hip.h) as ONE launch: a vector step of PPO on
// the Atari shape (configs/ppo/atari.yaml, 8 envs) was ~15 launches of 5-15 us each; behind the convolutions and the 6 400 -> 512
// product came the split-K epilogue, the heads' skinny product, xrl_policy_sample, the provider, xrl_rollout_poststep and the copy of
// the observations into their buffer slot -- five of them small launches whose work is a few thousand operations.  Here:
//   workgroup m < M  (one per row of the policy batch) sums its row of the split-K partials (splitk_epilogue_kernel's order: ws[0] + ws[1]
//                    + ...; + bias; activation) into LDS, forms logits and value with skinny_fwd_kernel's arithmetic (one wavefront per
//                    head: lane l takes k = l, l + 64, ... as an fma chain, 64-lane butterfly, + bias) and -- rows [0, n) -- samples
//                    as policy_sample_kernel does; rows [n, 2 n) store the value that bootstraps the previous step's cut paths;
//   workgroup M      (post_n > 0) poststep_body of the PREVIOUS vector step (reads what the provider left, writes that step's slots);
//   the others       copy the frames the policy acted on into memory.observations[t].
// Same numbers as the launches it replaces, bit for bit (tests/test_gpu_agent.py).  Reference: on_policy.py:128-169 (get_actions),
// ppo_agent.py:128,144-157 (store, bookkeeping), cnn.py:53-102 / actor_head.py / critic_head.py (the heads).
#include "common.h"
#include "rng.h"
#include "poststep.h"

namespace xrl {

constexpr int PT_THREADS = 1024;
constexpr int PT_MAX_ROWS = 64, PT_MAX_OUT = 17;        // rows of the policy batch, actions + the value

__global__ void __launch_bounds__(PT_THREADS) ppo_act_tail_kernel(xrl_ppo_act_tail_t p) {
    __shared__ __attribute__((aligned(16))) float h[1024 + 4];          // the row's hidden activations
    __shared__ unsigned long long ended_mask[64];
    __shared__ float s_heads[PT_MAX_OUT + 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.M, H = p.H, A = p.A, n = p.n;
    const int n_post = p.post_n > 0 ? 1 : 0;
    // workgroups [0, M): one per row of the policy batch (a row's heads, sample and stores need nothing of another row; one workgroup for
    // all rows pulled the whole 512 KB of split-K partials through ONE CU: 25 us); then the bookkeeping workgroup, then the copies
    if ((int)blockIdx.x == M && n_post) { poststep_body<PT_THREADS>(p.post, ended_mask); return; }
    if ((int)blockIdx.x >= M) {                                         // memory.observations[t] = obs: 16-byte copies
        const int nb = (int)gridDim.x - M - n_post, b = (int)blockIdx.x - M - n_post;
        const int64_t n16 = p.copy_bytes / 16;
        const uint4* src = reinterpret_cast<const uint4*>(p.copy_src);
        uint4* dst = reinterpret_cast<uint4*>(p.copy_dst);
        for (int64_t i = (int64_t)b * PT_THREADS + tid; i < n16; i += (int64_t)nb * PT_THREADS) dst[i] = src[i];
        return;
    }
    const int m = blockIdx.x;
    if (m >= n && !(p.bootv_prev && m < 2 * n)) return;                  // (a row nobody reads)
    // ---- the hidden layer: C[m, c] = act(sum_s ws[s][m][c] + bias[c])  (splitk_epilogue_kernel, element by element)
    {
        const int64_t total = (int64_t)M * H;
        for (int c = tid; c < H; c += PT_THREADS) {
            const int64_t i = (int64_t)m * H + c;
            // (the partials are requested eight at a time and added in order: one load per loop trip made the sum a chain of ks memory
            //  round trips -- 10.9 us for the launch)
            float s = p.ws[i];
            int q = 1;
            for (; q + 8 <= p.ks; q += 8) {
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w8[j] = p.ws[(size_t)(q + j) * total + i];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += w8[j];
            }
            for (; q < p.ks; ++q) s += p.ws[(size_t)q * total + i];
            const float z = s + (p.bias ? p.bias[c] : 0.f);
            float y = z;
            XRL_ACT_DISPATCH(p.act, y = act_apply_c<ACT>(z);)
            h[c] = y;
        }
    }
    __syncthreads();
    // ---- logits (wave 0) and value (wave 1): skinny_fwd_kernel's statements with A = the row in LDS
    if (wave < 2) {
        const int g = wave;
        const float* a = h;
        const float* B = g ? p.w_critic : p.w_actor;
        const float* bias = g ? p.b_critic : p.b_actor;
        const int N = g ? 1 : A;
        constexpr int SN = 16;
        float acc[SN];
#pragma unroll
        for (int j = 0; j < SN; ++j) acc[j] = 0.f;
        int k = lane;
        for (; k + 64 * 3 < H; k += 64 * 4) {
            float av[4], bv[4][SN];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = a[k + 64 * u];
#pragma unroll
                for (int j = 0; j < SN; ++j) bv[u][j] = j < N ? B[(size_t)j * H + k + 64 * u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < SN; ++j) acc[j] = fmaf(av[u], bv[u][j], acc[j]);
        }
        for (; k < H; k += 64) {
            const float av = a[k];
#pragma unroll
            for (int j = 0; j < SN; ++j)
                if (j < N) acc[j] = fmaf(av, B[(size_t)j * H + k], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < SN; ++j) {
            if (j >= N) break;
            const float s = wave_sum(acc[j]);
            if (lane == 0) s_heads[(g ? A : 0) + j] = s + (bias ? bias[j] : 0.f);
        }
    }
    __syncthreads();
    if (p.heads_out && tid <= A) p.heads_out[(size_t)m * (A + 1) + tid] = s_heads[tid];
    if (tid != 0) return;
    // ---- policy_sample_kernel, categorical: rows [0, n) act; rows [n, 2 n): the value that bootstraps the previous step's cut paths
    const float* hd = s_heads;
    if (m >= n) { p.bootv_prev[m - n] = hd[A]; return; }
    if (!p.act_out) return;
    const int e = m;
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    float u;
    if (p.noise) u = p.noise[e];
    else { uint32_t r[4]; philox4x32(p.seed, (uint32_t)e, step, STREAM_ACTION, r); u = u01(r[0]); }
    float mx = hd[0];
    for (int j = 1; j < A; ++j) mx = fmaxf(mx, hd[j]);
    float se = 0.f;
    for (int j = 0; j < A; ++j) se += expf(hd[j] - mx);
    const float lse = mx + logf(se);
    int a = A - 1;
    float c = 0.f;
    for (int j = 0; j < A; ++j) {
        c += expf(hd[j] - lse);
        if (c > u) { a = j; break; }
    }
    p.act_out[e] = (float)a;
    if (p.env_action) p.env_action[e] = a;
    if (p.val_out) p.val_out[e] = hd[A];
    p.logp_out[e] = hd[a] - lse;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_ppo_act_tail(const xrl_ppo_act_tail_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_ppo_act_tail_t& p = *pp;
    XRL_CHECK_ARG(p.ws && p.ks >= 1 && p.M >= 1 && p.M <= PT_MAX_ROWS && p.H >= 64 && p.H <= 1024 && (p.H & 63) == 0);
    XRL_CHECK_ARG(p.w_actor && p.w_critic && p.A >= 1 && p.A <= PT_MAX_OUT - 1 && p.n >= 1 && p.n <= PT_THREADS);
    XRL_CHECK_ARG(p.M >= p.n && (p.bootv_prev == nullptr || p.M >= 2 * p.n));
    XRL_CHECK_ARG((p.act_out && p.logp_out) || (!p.act_out && p.bootv_prev));
    XRL_CHECK_ARG(p.copy_bytes >= 0 && (p.copy_bytes & 15) == 0 &&
                  (p.copy_bytes == 0 || (p.copy_src && p.copy_dst && ((reinterpret_cast<uintptr_t>(p.copy_src) | reinterpret_cast<uintptr_t>(p.copy_dst)) & 15) == 0)));
    if (p.post_n > 0) {
        const xrl_poststep_t& q = p.post;
        XRL_CHECK_ARG(q.n == p.post_n && q.reward && q.terminated && q.truncated && q.rew_out && q.term_out && q.seg_out && q.ret_track &&
                      q.ret_mean && q.ret_var && q.ret_count && q.D > 0);
        XRL_CHECK_ARG(!q.use_obsnorm || (q.obs_mean && q.obs_var));
        XRL_CHECK_ARG(q.next_obs_norm == nullptr || q.next_obs != nullptr);
    }
    int n_copy = 0;
    if (p.copy_bytes > 0) { n_copy = (int)((p.copy_bytes / 16 + 4 * PT_THREADS - 1) / (4 * PT_THREADS)); n_copy = n_copy < 1 ? 1 : (n_copy > 32 ? 32 : n_copy); }
    hipLaunchKernelGGL(ppo_act_tail_kernel, dim3(p.M + (p.post_n > 0 ? 1 : 0) + n_copy), dim3(PT_THREADS), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
