// TD-target / value-factorisation kernels for the off-policy learners.
//   dqn_td_kernel   : xuance/torch/learners/qlearning_family/dqn_learner.py:39-46 (+ DDQN rule ddqn_learner.py:39-47)
//   qmix_kernel     : multi_agent_rl/qmix_learner.py:34-86, iql_learner.py:63-81, rl_models/heads/q_mix_head.py:66-95
//   sync_target     : dqn_learner.py:56-57 / qmix_learner.py:105-106 (hard target copy), graph-capturable
// All HBM/latency bound: per row a few hundred bytes; the dense work (Q-networks, hyper-networks) runs in gemm.hip.
#include "common.h"
#include "rng.h"

namespace xrl {

// Loss of one TD error and its derivative: nn.MSELoss (delta <= 0: dqn_learner.py:25,46) or nn.HuberLoss(delta) (the form the
// reference's learners use where they switch it on: learners/base/marl_learner.py:193-197 `use_huber_loss` / `huber_delta`;
// torch: 0.5 z^2 for |z| < delta, delta (|z| - 0.5 delta) beyond; backward z, or +-delta).  Both with reduction "mean": the
// caller divides by the row count.
__device__ __forceinline__ double td_loss(float td, float delta, float& dl) {
    if (!(delta > 0.f)) { dl = 2.f * td; return (double)td * td; }
    const float a = fabsf(td);
    if (a < delta) { dl = td; return (double)(0.5f * td * td); }
    dl = copysignf(delta, td);
    return (double)(delta * (a - 0.5f * delta));
}

__global__ void __launch_bounds__(256) dqn_td_kernel(xrl_dqn_td_t p) {
    __shared__ double scratch[16];
    const int chunk = (p.M + p.n_split - 1) / p.n_split;
    const int mbeg = blockIdx.x * chunk, mend = min(p.M, mbeg + chunk);
    const float invM = 1.f / (float)p.M;
    double acc_l = 0.0, acc_q = 0.0;
    const int A = p.A;
    const bool duel = p.dueling != 0;
    // Q(s, j) of a head row.  Plain head: r[j].  Dueling head (DuelingQValueHead.forward, q_head.py:75-77): the row is
    // [advantages (A) | value], Q = V + (A_j - mean(A)).
    auto row_mean = [&](const float* r) { float s = 0.f; for (int j = 0; j < A; ++j) s += r[j]; return s / (float)A; };
    for (int m = mbeg + threadIdx.x; m < mend; m += blockDim.x) {
        const float* qe = p.q_eval + (size_t)m * p.ld;
        const float* qn = p.q_next + (size_t)m * p.ld;
        const int a = (int)p.actions[m];
        const float me = duel ? row_mean(qe) : 0.f, mn = duel ? row_mean(qn) : 0.f;
        const float pred = duel ? qe[A] + (qe[a] - me) : qe[a];          // :42
        float tq;
        if (p.q_next_eval) {                                             // double-Q: argmax of the eval net
            const float* qs = p.q_next_eval + (size_t)m * p.ld;          // (argmax over advantages == argmax over Q)
            int best = 0; float bv = qs[0];
            for (int j = 1; j < A; ++j) if (qs[j] > bv) { bv = qs[j]; best = j; }
            tq = duel ? qn[A] + (qn[best] - mn) : qn[best];
        } else {
            tq = duel ? qn[A] + (qn[0] - mn) : qn[0];
            for (int j = 1; j < A; ++j) tq = fmaxf(tq, duel ? qn[A] + (qn[j] - mn) : qn[j]);   // :43
        }
        const float y = p.rewards[m] + p.gamma * (1.f - p.terminals[m]) * tq;    // :44
        const float td = pred - y;
        float* dq = p.d_q + (size_t)m * p.ld;
        float dl;
        const double lo = td_loss(td, p.huber_delta, dl);
        const float g = dl * invM;                                       // MSELoss / HuberLoss backward through gather
        if (duel) {      // dA_j = g [j == a] + (-g) / A (backward of `- mean`), dV = g
            for (int j = 0; j < A; ++j) dq[j] = ((j == a) ? g : 0.f) + (-g) / (float)A;
            dq[A] = g;
        } else {
            for (int j = 0; j < A; ++j) dq[j] = (j == a) ? g : 0.f;
        }
        if (p.diag) { p.diag[m] = pred; p.diag[p.M + m] = y; }
        acc_l += lo; acc_q += pred;
    }
    const double t0 = block_sum(acc_l, scratch), t1 = block_sum(acc_q, scratch);
    if (threadIdx.x == 0) {
        double* q = p.partials + (size_t)blockIdx.x * 8;
        q[0] = t0; q[1] = t1;
        for (int j = 2; j < 8; ++j) q[j] = 0.0;
    }
}

// Q layer + TD target + the Q layer's data gradient as ONE launch (dqn_learner.py:39-46, ddqn_learner.py:39-47 behind a
// BasicQhead, q_head.py:8-39): at batch 32 the 512 -> n_actions layer, the TD rule and the gradient back into the hidden layer were
// three launches of 4.5-7 us for a few thousand multiply-adds each.  One workgroup per transition m: the (row, action) dot
// products of its up to three rows -- hidden activations of the eval network on obs[m] (and on next_obs[m] under double-Q) and of
// the target network on next_obs[m] -- are spread over the four waves, the TD rule is dqn_td_kernel's, and
// d_h[m][j] = dQ[m][a_m] * W[a_m][j] * act'(h[m][j]) leaves through all 256 threads.  The Q values go where the layered path puts
// them (callbacks read them there), d_q feeds the layer's weight-gradient GEMM as before.
__global__ void __launch_bounds__(256) dqn_head_td_kernel(xrl_dqn_head_td_t p) {
    __shared__ float s_q[3 * 64];
    const int A = p.A, H = p.H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_rows = p.double_q ? 3 : 2;
    for (int m = blockIdx.x; m < p.M; m += gridDim.x) {
        const float* h0 = p.h_eval + (size_t)m * p.ld_h;
        const float* h1 = p.h_target + (size_t)m * p.ld_h;
        const float* h2 = p.h_eval + (size_t)(p.M + m) * p.ld_h;
        // everything the tail needs that does not depend on the Q values is requested NOW (the taken action, its weight row and the
        // hidden activations for d_h): the launch is a chain of memory round trips otherwise (three of them: 9 us)
        const int a_taken = min(max((int)p.actions[m], 0), p.A - 1);   // (a corrupt stored action must not index outside the row)
        const float rew = p.rewards[m], ter = p.terminals[m];
        const float* wa = p.w_eval + (size_t)a_taken * H;
        float pw[2], ph[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = threadIdx.x + 256 * u;
            pw[u] = j < H ? wa[j] : 0.f;
            ph[u] = j < H ? h0[j] : 0.f;
        }
        for (int pr = wave; pr < n_rows * A; pr += 4) {
            const int r = pr / A, a = pr - r * A;
            const float* h = r == 0 ? h0 : (r == 1 ? h1 : h2);
            const float* w = (r == 1 ? p.w_target : p.w_eval) + (size_t)a * H;
            float acc = 0.f;
            for (int k = lane; k < H; k += 64) acc = fmaf(h[k], w[k], acc);
            acc = wave_sum(acc);
            if (lane == 0) s_q[r * 64 + a] = acc + (r == 1 ? p.b_target : p.b_eval)[a];
        }
        __syncthreads();
        if (threadIdx.x < A) {                                           // the layered path's output level
            p.q_eval[(size_t)m * p.ld_q + threadIdx.x] = s_q[threadIdx.x];
            p.q_target[(size_t)m * p.ld_q + threadIdx.x] = s_q[64 + threadIdx.x];
            if (p.double_q) p.q_eval[(size_t)(p.M + m) * p.ld_q + threadIdx.x] = s_q[128 + threadIdx.x];
        }
        // (every thread evaluates the TD rule on the LDS values: no second barrier, no broadcast)
        const float pred = s_q[a_taken];                                 // :42
        float tq;
        if (p.double_q) {                                                // argmax of the eval net on next_obs
            int best = 0; float bv = s_q[128];
            for (int j = 1; j < A; ++j) if (s_q[128 + j] > bv) { bv = s_q[128 + j]; best = j; }
            tq = s_q[64 + best];
        } else {
            tq = s_q[64];
            for (int j = 1; j < A; ++j) tq = fmaxf(tq, s_q[64 + j]);     // :43
        }
        const float y = rew + p.gamma * (1.f - ter) * tq;                // :44
        const float td = pred - y;
        float dl;
        const double lo = td_loss(td, p.huber_delta, dl);
        const float g = dl / (float)p.M;                                 // MSELoss / HuberLoss backward through gather
        if (threadIdx.x < A) p.d_q[(size_t)m * p.ld_q + threadIdx.x] = ((int)threadIdx.x == a_taken) ? g : 0.f;
        if (threadIdx.x == 0) {
            if (p.diag) { p.diag[m] = pred; p.diag[p.M + m] = y; }
            double* q = p.partials + (size_t)m * 8;
            q[0] = lo; q[1] = pred;
            for (int j = 2; j < 8; ++j) q[j] = 0.0;
        }
        float* dh = p.d_h + (size_t)m * p.ld_h;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = threadIdx.x + 256 * u;
            if (j < H) dh[j] = g * pw[u] * act_grad_from_out(ph[u], p.act);
        }
        for (int j = threadIdx.x + 512; j < H; j += 256) dh[j] = g * wa[j] * act_grad_from_out(h0[j], p.act);
        __syncthreads();                                                 // (s_q is rewritten by the next transition)
    }
}

// Everything between the last convolution and the convolution stack's backward pass of a DQN update on Basic_CNN features
// (cnn.py:11-50 -> q_head.py:8-39 -> dqn_learner.py:39-46), ONE launch, one workgroup per transition m:
//   global max-pool of the three frames the transition needs (eval(obs), target(next_obs), eval(next_obs) under double-Q)
//   -> hidden layer F -> H of both networks -> Q layer -> TD rule, loss terms, dQ -> d_h -> d_feat = d_h . W1
//   -> the pool's backward: dY of the last convolution (the gradient lands on the first maximum, times relu').
// At batch 32 these were six launches (pool 4.8, grouped dense GEMM 9.0, xrl_dqn_head_td 9.7, backward-data 8.9, pool backward 4.8 us
// + their boundaries) for 0.2 MFLOP per transition.  Lane mapping of the two F x H products: 16 lanes share a weight row (one
// 256-byte line, a float4 each), a wave covers 4 rows per load instruction (1 KB contiguous), the workgroup's waves 4 rows each per round.  The eval matrix is loaded ONCE, all rounds up front (H / 16 float4 per lane), and stays in registers for the
// backward product; the forward partial sums meet through a quad reduction (DPP) + LDS, the backward ones (one float4 of
// d_feat per lane, summed over the lane's rows) through LDS in a fixed order.  512 threads: 8 waves share the rounds (16 float4 per
// matrix and lane, both matrices requested at the start), the pool's positions and the Q layer's (row, action) pairs.
constexpr int TAIL_F = 64, TAIL_HMAX = 512, TAIL_T = 512, TAIL_W = TAIL_T / 64, TAIL_RR = 4 * TAIL_W;   // weight rows per round
constexpr int TAIL_RMAX = TAIL_HMAX / TAIL_RR, TAIL_PQ = 16;                     // up to TAIL_W * TAIL_PQ = 128 pooled positions
constexpr int TAIL_QP = 2;                                                        // (row, action) pairs of the Q layer per wave

__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

__global__ void __launch_bounds__(TAIL_T) dqn_tail_td_kernel(xrl_dqn_tail_td_t p) {
    __shared__ __attribute__((aligned(16))) float s_f[3][TAIL_F];
    __shared__ __attribute__((aligned(16))) float s_part[3][TAIL_HMAX][4];
    __shared__ float s_h[3][TAIL_HMAX];
    __shared__ float s_dh[TAIL_HMAX];
    __shared__ float s_q[3 * 64];
    __shared__ float s_mv[3][TAIL_W][TAIL_F];
    __shared__ int s_mi[3][TAIL_W][TAIL_F];
    __shared__ int s_arg[TAIL_F];
    __shared__ __attribute__((aligned(16))) float s_red[TAIL_RR][TAIL_F];
    __shared__ float s_df[TAIL_F];
    constexpr int F = TAIL_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, H = p.H, P = p.P, M = p.M, m = blockIdx.x;
    const int n_rows = p.double_q ? 3 : 2, rounds = (H + TAIL_RR - 1) / TAIL_RR;
    const int k4 = lane & 15, rsub = lane >> 4, jrow = 4 * wave + rsub;            // this lane's weight row within a round
    // ---- every load that does not depend on computed values, now: both hidden matrices (all rounds; the eval one stays in
    // registers for the backward product), the three frames' activations, the Q-layer rows this wave will need, the scalars
    float4 we[TAIL_RMAX], wt[TAIL_RMAX];
#pragma unroll
    for (int i = 0; i < TAIL_RMAX; ++i) {
        const int j = min(TAIL_RR * i + jrow, H - 1);                              // (rows beyond H: a valid address, unused)
        we[i] = *reinterpret_cast<const float4*>(p.w1_eval + (size_t)j * F + 4 * k4);
        wt[i] = *reinterpret_cast<const float4*>(p.w1_target + (size_t)j * F + 4 * k4);
    }
    const float* frame[3] = {p.y_eval + (size_t)m * P * F, p.y_target + (size_t)m * P * F, p.y_eval + (size_t)(p.double_q ? M + m : m) * P * F};
    const int f = lane, pg = wave;                                                 // pool: thread (position group, filter)
    float pv[3][TAIL_PQ];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < TAIL_PQ; ++q) pv[r][q] = frame[r][(size_t)min(pg + TAIL_W * q, P - 1) * F + f];
    const int a_taken = min(max((int)p.actions[m], 0), p.A - 1);
    const float rew = p.rewards[m], ter = p.terminals[m];
    float w2[TAIL_QP][TAIL_HMAX / 64];                                             // Q-layer rows of this wave's (row, action) pairs
#pragma unroll
    for (int u = 0; u < TAIL_QP; ++u) {
        const int pr = min(wave + TAIL_W * u, n_rows * A - 1), r = pr / A, a = pr - r * A;
        const float* w = (r == 1 ? p.w2_target : p.w2_eval) + (size_t)a * H;
#pragma unroll
        for (int t = 0; t < TAIL_HMAX / 64; ++t) w2[u][t] = w[min(lane + 64 * t, H - 1)];
    }
    const float wa_j = p.w2_eval[(size_t)a_taken * H + min(tid, H - 1)];           // (the taken action's Q row, for d_h)
    const float b1e = p.b1_eval[min(tid, H - 1)], b1t = p.b1_target[min(tid, H - 1)];
    // ---- global max-pool, first maximum wins (xrl_maxpool_hw_fwd's rule: larger value, then smaller position)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < TAIL_PQ; ++q) {
            const int pos = pg + TAIL_W * q;
            if (pos < P && pv[r][q] > best) { best = pv[r][q]; bi = pos; }
        }
        s_mv[r][pg][f] = best; s_mi[r][pg][f] = bi;
    }
    __syncthreads();
    if (tid < 3 * F) {
        const int r = tid >> 6;
        float best = s_mv[r][0][f];
        int bi = s_mi[r][0][f];
#pragma unroll
        for (int g = 1; g < TAIL_W; ++g) {
            const float v = s_mv[r][g][f];
            const int vi = s_mi[r][g][f];
            if (v > best || (v == best && vi < bi)) { best = v; bi = vi; }
        }
        s_f[r][f] = best;
        if (r == 0) { s_arg[f] = bi; p.feat_eval[(size_t)m * p.ld_f + f] = best; if (p.arg) p.arg[(size_t)m * F + f] = bi; }
        if (r == 1 && p.feat_target) p.feat_target[(size_t)m * p.ld_f + f] = best;
        if (r == 2 && p.double_q) p.feat_eval[(size_t)(M + m) * p.ld_f + f] = best;
    }
    __syncthreads();
    // ---- hidden layer: partial dot products per (row, quad of k), eval network on rows 0 (and 2), target network on row 1
    const float4 f0 = *reinterpret_cast<const float4*>(&s_f[0][4 * k4]), f1 = *reinterpret_cast<const float4*>(&s_f[1][4 * k4]),
                 f2 = *reinterpret_cast<const float4*>(&s_f[2][4 * k4]);
#pragma unroll
    for (int i = 0; i < TAIL_RMAX; ++i) {
        const int j = TAIL_RR * i + jrow;
        if (j < H) {                                                               // (uniform per 16 lanes: the quad reduction is safe)
            const float s0 = quad_sum(dot4(f0, we[i])), s1 = quad_sum(dot4(f1, wt[i]));
            if ((k4 & 3) == 0) { s_part[0][j][k4 >> 2] = s0; s_part[1][j][k4 >> 2] = s1; }
            if (p.double_q) {
                const float s2 = quad_sum(dot4(f2, we[i]));
                if ((k4 & 3) == 0) s_part[2][j][k4 >> 2] = s2;
            }
        }
    }
    __syncthreads();
    if (tid < H) {
        for (int r = 0; r < n_rows; ++r) {
            const float4 q4 = *reinterpret_cast<const float4*>(&s_part[r][tid][0]);
            const float hv = act_apply(((q4.x + q4.y) + (q4.z + q4.w)) + (r == 1 ? b1t : b1e), p.act);
            s_h[r][tid] = hv;
            if (r == 0) p.h_eval[(size_t)m * p.ld_h + tid] = hv;
            if (r == 2) p.h_eval[(size_t)(M + m) * p.ld_h + tid] = hv;
        }
    }
    __syncthreads();
    // ---- Q layer ((row, action) pairs over the waves), TD rule: xrl_dqn_head_td's statements on the LDS rows
#pragma unroll
    for (int u = 0; u < TAIL_QP; ++u) {
        const int pr = wave + TAIL_W * u;
        if (pr < n_rows * A) {
            const int r = pr / A, a = pr - r * A;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < TAIL_HMAX / 64; ++t) if (lane + 64 * t < H) acc = fmaf(s_h[r][lane + 64 * t], w2[u][t], acc);
            acc = wave_sum(acc);
            if (lane == 0) s_q[r * 64 + a] = acc + (r == 1 ? p.b2_target : p.b2_eval)[a];
        }
    }
    for (int pr = wave + TAIL_W * TAIL_QP; pr < n_rows * A; pr += TAIL_W) {         // (more pairs than prefetched: n_actions > 5)
        const int r = pr / A, a = pr - r * A;
        const float* w = (r == 1 ? p.w2_target : p.w2_eval) + (size_t)a * H;
        float acc = 0.f;
        for (int k = lane; k < H; k += 64) acc = fmaf(s_h[r][k], w[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) s_q[r * 64 + a] = acc + (r == 1 ? p.b2_target : p.b2_eval)[a];
    }
    __syncthreads();
    if (tid < A) {
        p.q_eval[(size_t)m * p.ld_q + tid] = s_q[tid];
        p.q_target[(size_t)m * p.ld_q + tid] = s_q[64 + tid];
        if (p.double_q) p.q_eval[(size_t)(M + m) * p.ld_q + tid] = s_q[128 + tid];
    }
    const float pred = s_q[a_taken];                                       // dqn_learner.py:42
    float tq;
    if (p.double_q) {                                                      // ddqn_learner.py:40-44: argmax of the eval net on next_obs
        int best = 0; float bv = s_q[128];
        for (int j = 1; j < A; ++j) if (s_q[128 + j] > bv) { bv = s_q[128 + j]; best = j; }
        tq = s_q[64 + best];
    } else {
        tq = s_q[64];
        for (int j = 1; j < A; ++j) tq = fmaxf(tq, s_q[64 + j]);           // :43
    }
    const float y = rew + p.gamma * (1.f - ter) * tq;                      // :44
    const float td = pred - y;
    float dl;
    const double lo = td_loss(td, p.huber_delta, dl);
    const float g = dl / (float)M;                                         // MSELoss / HuberLoss backward through gather
    if (tid < A) p.d_q[(size_t)m * p.ld_q + tid] = (tid == a_taken) ? g : 0.f;
    if (tid == 0) {
        if (p.diag) { p.diag[m] = pred; p.diag[M + m] = y; }
        double* q = p.partials + (size_t)m * 8;
        q[0] = lo; q[1] = pred;
        for (int j = 2; j < 8; ++j) q[j] = 0.0;
    }
    if (tid < H) {
        const float d = g * wa_j * act_grad_from_out(s_h[0][tid], p.act);
        s_dh[tid] = d;
        p.d_h[(size_t)m * p.ld_h + tid] = d;
    }
    __syncthreads();
    // ---- d_feat = d_h . W1 from the registers (and this transition's term of the dense layers' gradients into ITS slab:
    // rank-1 products, written where the weight-gradient GEMM would have put their sum), then the pool's backward
    float* slab = p.slabs ? p.slabs + (size_t)m * p.slab_stride : nullptr;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TAIL_RMAX; ++i) {
        const int j = TAIL_RR * i + jrow;
        if (j < H) {
            const float d = s_dh[j];
            acc.x = fmaf(d, we[i].x, acc.x); acc.y = fmaf(d, we[i].y, acc.y); acc.z = fmaf(d, we[i].z, acc.z); acc.w = fmaf(d, we[i].w, acc.w);
            if (slab) *reinterpret_cast<float4*>(slab + p.off_w1 + (size_t)j * F + 4 * k4) = make_float4(d * f0.x, d * f0.y, d * f0.z, d * f0.w);
        }
    }
    if (slab) {
        if (tid < H) {
            slab[p.off_b1 + tid] = s_dh[tid];
            const float gh = g * s_h[0][tid];
            for (int a = 0; a < A; ++a) slab[p.off_w2 + (size_t)a * H + tid] = (a == a_taken) ? gh : 0.f;
        }
        if (tid < A) slab[p.off_b2 + tid] = (tid == a_taken) ? g : 0.f;
    }
    *reinterpret_cast<float4*>(&s_red[jrow][4 * k4]) = acc;
    __syncthreads();
    if (tid < F) {
        float sum = s_red[0][tid];
#pragma unroll
        for (int g2 = 1; g2 < TAIL_RR; ++g2) sum += s_red[g2][tid];
        s_df[tid] = sum;
        if (p.d_feat) p.d_feat[(size_t)m * p.ld_f + tid] = sum;
    }
    __syncthreads();
    float* dy = p.dy + (size_t)m * P * F;
    for (int i = tid; i < P * F; i += TAIL_T) {
        const int q = i >> 6, ff = i & 63;
        dy[i] = (q == s_arg[ff] && s_f[0][ff] > 0.f) ? s_df[ff] : 0.f;    // xrl_maxpool_hw_bwd: first maximum, times relu'
    }
}

// The acting twin of dqn_tail_td_kernel (off_policy.py:129-148 on deep_q_network.py:61-80): pool of env e's frame, hidden layer,
// Q layer, greedy action + the per-env epsilon coin (xrl_egreedy's rule and Philox stream) -- one launch instead of four.
__global__ void __launch_bounds__(TAIL_T) dqn_act_tail_kernel(xrl_dqn_act_tail_t p) {
    __shared__ __attribute__((aligned(16))) float s_f[TAIL_F];
    __shared__ __attribute__((aligned(16))) float s_part[TAIL_HMAX][4];
    __shared__ float s_h[TAIL_HMAX];
    __shared__ float s_q[64];
    __shared__ float s_mv[TAIL_W][TAIL_F];
    constexpr int F = TAIL_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, H = p.H, P = p.P, e = blockIdx.x;
    const int k4 = lane & 15, rsub = lane >> 4, jrow = 4 * wave + rsub;
    float4 we[TAIL_RMAX];
#pragma unroll
    for (int i = 0; i < TAIL_RMAX; ++i)
        we[i] = *reinterpret_cast<const float4*>(p.w1 + (size_t)min(TAIL_RR * i + jrow, H - 1) * F + 4 * k4);
    const float* frame = p.y + (size_t)e * P * F;
    float pv[TAIL_PQ];
#pragma unroll
    for (int q = 0; q < TAIL_PQ; ++q) pv[q] = frame[(size_t)min(wave + TAIL_W * q, P - 1) * F + lane];
    float w2[TAIL_QP][TAIL_HMAX / 64];
#pragma unroll
    for (int u = 0; u < TAIL_QP; ++u) {
        const float* w = p.w2 + (size_t)min(wave + TAIL_W * u, A - 1) * H;
#pragma unroll
        for (int t = 0; t < TAIL_HMAX / 64; ++t) w2[u][t] = w[min(lane + 64 * t, H - 1)];
    }
    const float b1 = p.b1[min(tid, H - 1)];
    const uint32_t step = p.step + (p.step_dev ? *p.step_dev : 0u);
    float eps = p.eps_dev ? *p.eps_dev : p.eps;
    if (p.eps_sched) {                                  // the host's schedule (off_policy.py:119-127), evaluated from the step counter
        const long long cs = (long long)min(step, p.eps_kstar) * (long long)p.eps_n;          // current_step
        eps = (float)__dsub_rn(p.eps_start, __dmul_rn((double)cs, p.eps_delta));
    }
    float best = -INFINITY;
#pragma unroll
    for (int q = 0; q < TAIL_PQ; ++q) if (wave + TAIL_W * q < P) best = fmaxf(best, pv[q]);
    s_mv[wave][lane] = best;
    __syncthreads();
    if (tid < F) {
        float v = s_mv[0][tid];
#pragma unroll
        for (int g = 1; g < TAIL_W; ++g) v = fmaxf(v, s_mv[g][tid]);
        s_f[tid] = v;
        if (p.feat) p.feat[(size_t)e * p.ld_f + tid] = v;
    }
    __syncthreads();
    const float4 f0 = *reinterpret_cast<const float4*>(&s_f[4 * k4]);
#pragma unroll
    for (int i = 0; i < TAIL_RMAX; ++i) {
        const int j = TAIL_RR * i + jrow;
        if (j < H) {
            const float s0 = quad_sum(dot4(f0, we[i]));
            if ((k4 & 3) == 0) s_part[j][k4 >> 2] = s0;
        }
    }
    __syncthreads();
    if (tid < H) {
        const float4 q4 = *reinterpret_cast<const float4*>(&s_part[tid][0]);
        s_h[tid] = act_apply(((q4.x + q4.y) + (q4.z + q4.w)) + b1, p.act);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TAIL_QP; ++u) {
        const int a = wave + TAIL_W * u;
        if (a < A) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < TAIL_HMAX / 64; ++t) if (lane + 64 * t < H) acc = fmaf(s_h[lane + 64 * t], w2[u][t], acc);
            acc = wave_sum(acc);
            if (lane == 0) s_q[a] = acc + p.b2[a];
        }
    }
    for (int a = wave + TAIL_W * TAIL_QP; a < A; a += TAIL_W) {
        const float* w = p.w2 + (size_t)a * H;
        float acc = 0.f;
        for (int k = lane; k < H; k += 64) acc = fmaf(s_h[k], w[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) s_q[a] = acc + p.b2[a];
    }
    __syncthreads();
    if (tid < A && p.q) p.q[(size_t)e * p.ld_q + tid] = s_q[tid];
    if (tid == 0) {
        int bi = 0;
        float bv = s_q[0];
        for (int j = 1; j < A; ++j) if (s_q[j] > bv) { bv = s_q[j]; bi = j; }   // argmax: first maximal index (xrl_egreedy)
        uint32_t r[4];
        philox4x32(p.seed, (uint32_t)e, step, STREAM_EGREEDY, r);
        const int a = (u01(r[0]) < eps) ? (int)(r[1] % (uint32_t)A) : bi;
        p.action[e] = a;
        if (p.action_f) p.action_f[e] = (float)a;
    }
}

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }

// One wavefront per batch row b.  Lane n < N owns agent n, lane h < H owns mixer hidden unit h.
__global__ void __launch_bounds__(64) qmix_kernel(xrl_qmix_t p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = p.N, A = p.A, H = p.H;
    float qe = 0.f, qn = 0.f, mask = 0.f;
    int a_taken = 0;
    // recurrent branch: step mask and its sum (every wave adds the B values in the same order)
    float fl = 1.f, inv_norm = 0.f;
    if (p.filled) {
        // every wave forms sum(filled) itself, in the same order; loads issued 8 at a time (a dependent chain of 30
        // round trips cost most of this kernel's 19 us at 1 920 rows)
        float s = 0.f;
        int i = lane;
        if (p.B <= 64 * 32) {
            // the step mask's sum as ONE round trip (32 clamped loads per lane in flight; a count of 0 / 1 flags: exact in any order) --
            // the loop below is four dependent round trips at the recurrent update's 1 920 rows, in every one of its 1 920 workgroups
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = p.filled[min(lane + 64 * u, p.B - 1)];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += lane + 64 * u < p.B ? v[u] : 0.f;
            i = p.B;
        }
        for (; i + 64 * 7 < p.B; i += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p.filled[i + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < p.B; i += 64) s += p.filled[i];
        inv_norm = 1.f / wave_sum(s);
        fl = p.filled[b];
    }
    float iql_msum = 0.f;
    if (p.mixer == 2) {                                   // IQL: sum(mask) over all B*N entries, one strided pass per wave
        float s = 0.f;
        for (int i = lane; i < p.B * N; i += 64) s += p.agent_mask[i];
        iql_msum = wave_sum(s);
    }
    if (lane < N) {
        const size_t row = (size_t)b * N + lane;
        mask = p.agent_mask[row] * fl;                                                // outputs.py:138-143
        a_taken = min(max((int)p.actions[row], 0), A - 1);
        qe = p.q_eval[row * p.ldq + a_taken] * mask;                                  // qmix_learner.py:48-50,60
        const float* qt = p.q_next + row * p.ldq;
        const float* av = p.avail_next ? p.avail_next + row * A : nullptr;
        if (p.double_q) {                                                             // :52-55, iql_learner.py:68-71
            const float* qs = p.q_next_eval + row * p.ldq;
            int best = 0; float bv = (av && av[0] == 0.f) ? -1e10f : qs[0];
            for (int j = 1; j < A; ++j) {
                const float v = (av && av[j] == 0.f) ? -1e10f : qs[j];                // value_factorization.py:87-90
                if (v > bv) { bv = v; best = j; }
            }
            qn = (av && av[best] == 0.f) ? -1e10f : qt[best];                         // iql_learner.py:75-81
        } else {                                                                      // :57-58
            qn = (av && av[0] == 0.f) ? -1e10f : qt[0];
            for (int j = 1; j < A; ++j) qn = fmaxf(qn, (av && av[j] == 0.f) ? -1e10f : qt[j]);
        }
        if (p.mixer == 2) {
            // IQL_Learner (iql_learner.py:98-117): per-agent TD on the UNMASKED taken values, mask applied to the error
            const float msum = iql_msum;
            const float y = p.rewards[row] + (1.f - (p.terminals[row] != 0.f ? 1.f : 0.f)) * p.gamma * qn;      // :113
            const float qraw = p.q_eval[row * p.ldq + a_taken];
            const float td = (qraw - y) * mask;                                                                  // :116
            float* dq = p.d_q + row * p.ldq;
            for (int j = 0; j < A; ++j) dq[j] = (j == a_taken) ? 2.f * td * mask / msum : 0.f;                   // :117
            qe = td; qn = qraw;
            if (p.diag) { p.diag[row] = qraw; p.diag[(size_t)p.B * N + row] = y; }
        } else {
            qn *= mask;                                                               // :61
        }
    }
    if (p.mixer == 2) {
        const float l = wave_sum(lane < N ? qe * qe : 0.f), q = wave_sum(lane < N ? qn : 0.f), m = wave_sum(mask);
        if (lane == 0) {
            double* o = p.partials + (size_t)b * 8;
            o[0] = l; o[1] = q; o[2] = m;                       // loss = sum o[0] / sum o[2]; predictQ = sum o[1] / (B N)
            for (int j = 3; j < 8; ++j) o[j] = 0.0;
        }
        return;
    }
    if (p.mixer == 1) {
        // VDN_Learner (vdn_learner.py:13-106): the mixer is the sum over agents (VDN_Mixer), no state, no parameters
        const float q_tot_e = wave_sum(qe), q_tot_n = wave_sum(qn);
        float r = 0.f, dn = 1.f;
        if (lane < N) { r = p.rewards[(size_t)b * N + lane]; dn = p.terminals[(size_t)b * N + lane] != 0.f ? 1.f : 0.f; }
        const float r_tot = wave_sum(r) / (float)N;
        float all_d = dn;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) all_d = fminf(all_d, __shfl_xor(all_d, off, 64));
        const float y = r_tot + (1.f - all_d) * p.gamma * q_tot_n;
        const float td = (q_tot_e - y) * fl;
        const float dq_tot = p.filled ? 2.f * td * fl * inv_norm : 2.f * td / (float)p.B;
        if (lane < N) {
            float* dq = p.d_q + ((size_t)b * N + lane) * p.ldq;
            for (int j = 0; j < A; ++j) dq[j] = (j == a_taken) ? dq_tot * mask : 0.f;
        }
        if (lane == 0) {
            double* q = p.partials + (size_t)b * 8;
            q[0] = (double)td * td; q[1] = q_tot_e; q[2] = p.filled ? (double)fl : 0.0;
            for (int j = 3; j < 8; ++j) q[j] = 0.0;
            if (p.diag) { p.diag[b] = q_tot_e; p.diag[p.B + b] = q_tot_n; p.diag[2 * (size_t)p.B + b] = y; }
        }
        return;
    }
    // mixing networks (q_mix_head.py:78-95)
    const float* e_raw = p.e_raw + (size_t)b * p.ld_e2;
    const float* t_raw = p.t_raw + (size_t)b * p.ld_t2;
    float pre_e = 0.f, pre_t = 0.f, w2e = 0.f, w2t = 0.f;
    if (lane < H) { pre_e = p.e_b1[(size_t)b * p.ld_e1 + lane]; pre_t = p.t_b1[(size_t)b * p.ld_t1 + lane]; }
    for (int n = 0; n < N; ++n) {
        const float qen = __shfl(qe, n, 64), qnn = __shfl(qn, n, 64);
        if (lane < H) {
            pre_e += qen * fabsf(e_raw[n * H + lane]);                                // bmm(agent_qs, |w1|) + b1
            pre_t += qnn * fabsf(t_raw[n * H + lane]);
        }
    }
    float hid_e = 0.f, hid_t = 0.f;
    if (lane < H) {
        hid_e = elu_f(pre_e); hid_t = elu_f(pre_t);
        w2e = fabsf(e_raw[N * H + lane]); w2t = fabsf(t_raw[N * H + lane]);
    }
    const float q_tot_e = wave_sum(hid_e * w2e) + e_raw[N * H + H];                   // bmm(hidden, |w2|) + b2
    const float q_tot_n = wave_sum(hid_t * w2t) + t_raw[N * H + H];
    // rewards_tot = mean over agents, terminals_tot = all agents terminated (qmix_learner.py:34-35)
    float r = 0.f, dn = 1.f;
    if (lane < N) { r = p.rewards[(size_t)b * N + lane]; dn = p.terminals[(size_t)b * N + lane] != 0.f ? 1.f : 0.f; }
    const float r_tot = wave_sum(r) / (float)N;
    float all_d = dn;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) all_d = fminf(all_d, __shfl_xor(all_d, off, 64));
    const float y = r_tot + (1.f - all_d) * p.gamma * q_tot_n;                        // :78
    const float td = (q_tot_e - y) * fl;                                              // :83
    // d mean(td^2) / d q_tot_eval (:86), or d (sum(td^2) / sum(filled)) (:84)
    const float dq_tot = p.filled ? 2.f * td * fl * inv_norm : 2.f * td / (float)p.B;
    // backward through the eval mixer
    float* d_raw = p.d_e_raw + (size_t)b * p.ld_e2;
    float d_pre = 0.f;
    if (lane < H) {
        const float w2_raw = e_raw[N * H + lane];
        const float sgn2 = (w2_raw > 0.f) - (w2_raw < 0.f);
        d_raw[N * H + lane] = dq_tot * hid_e * sgn2;                                  // through abs()
        d_pre = dq_tot * w2e * (pre_e > 0.f ? 1.f : expf(pre_e));                     // ELU'
        p.d_e_b1[(size_t)b * p.ld_e1 + lane] = d_pre;
    }
    if (lane == 0) d_raw[N * H + H] = dq_tot;
    for (int n = 0; n < N; ++n) {
        const float qen = __shfl(qe, n, 64);
        float contrib = 0.f;
        if (lane < H) {
            const float w1_raw = e_raw[n * H + lane];
            const float sgn1 = (w1_raw > 0.f) - (w1_raw < 0.f);
            d_raw[n * H + lane] = qen * d_pre * sgn1;
            contrib = d_pre * fabsf(w1_raw);
        }
        const float dqe = wave_sum(contrib);                                          // d loss / d (masked q_eval_n)
        if (lane == n) {
            float* dq = p.d_q + ((size_t)b * N + n) * p.ldq;
            for (int j = 0; j < A; ++j) dq[j] = (j == a_taken) ? dqe * mask : 0.f;
        }
    }
    if (lane == 0) {
        double* q = p.partials + (size_t)b * 8;
        q[0] = (double)td * td; q[1] = q_tot_e; q[2] = p.filled ? (double)fl : 0.0;
        for (int j = 3; j < 8; ++j) q[j] = 0.0;
        if (p.diag) { p.diag[b] = q_tot_e; p.diag[p.B + b] = q_tot_n; p.diag[2 * (size_t)p.B + b] = y; }
    }
}

__global__ void __launch_bounds__(256) sync_target_kernel(const float* __restrict__ params, float* __restrict__ target,
                                                          int64_t P, const xrl_adam_state_t* st, int freq) {
    if (st->step % freq != 0) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x)
        target[i] = params[i];
}

// Twin of qmix_kernel's monotonic-mixer branch with the SAME arithmetic, operation by operation, but every load that
// does not depend on another load issued up front (register arrays, N <= 8 agents, A <= 16 actions): the original walks
// a chain of ~10 dependent global round trips (step mask sum -> action -> Q[action] -> argmax inputs -> Q_target[best] ->
// hyper-network outputs -> rewards), which was most of its 8.5 us (32 rows) / 14 us (1 920 rows).
constexpr int QM_MAXN = 8, QM_MAXA = 16;
__global__ void __launch_bounds__(64) qmix_prefetch_kernel(xrl_qmix_t p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = p.N, A = p.A, H = p.H;
    // ---- loads, all independent of each other
    const int an = lane < N ? lane : N - 1;                   // per-agent loads happen in every lane (clamped): no divergence
    const size_t row = (size_t)b * N + an;
    const float l_mask = p.agent_mask[row], l_act = p.actions[row], l_rew = p.rewards[row], l_term = p.terminals[row];
    float qev[QM_MAXA], qtv[QM_MAXA], qsv[QM_MAXA], avv[QM_MAXA];
#pragma unroll
    for (int j = 0; j < QM_MAXA; ++j) {
        const int jj = j < A ? j : A - 1;
        qev[j] = p.q_eval[row * p.ldq + jj];
        qtv[j] = p.q_next[row * p.ldq + jj];
        qsv[j] = p.double_q ? p.q_next_eval[row * p.ldq + jj] : 0.f;
        avv[j] = p.avail_next ? p.avail_next[row * A + jj] : 1.f;
    }
    const int hl = lane < H ? lane : H - 1;
    const float* e_raw = p.e_raw + (size_t)b * p.ld_e2;
    const float* t_raw = p.t_raw + (size_t)b * p.ld_t2;
    float er[QM_MAXN], tr[QM_MAXN];
#pragma unroll
    for (int n = 0; n < QM_MAXN; ++n) {
        const int nn = n < N ? n : N - 1;
        er[n] = e_raw[nn * H + hl];
        tr[n] = t_raw[nn * H + hl];
    }
    const float l_eb1 = p.e_b1[(size_t)b * p.ld_e1 + hl], l_tb1 = p.t_b1[(size_t)b * p.ld_t1 + hl];
    const float l_ew2 = e_raw[N * H + hl], l_tw2 = t_raw[N * H + hl];
    const float l_eb2 = e_raw[N * H + H], l_tb2 = t_raw[N * H + H];
    float fl = 1.f, inv_norm = 0.f;
    if (p.filled) {
        float s = 0.f;
        int i = lane;
        if (p.B <= 64 * 32) {
            // the step mask's sum as ONE round trip (32 clamped loads per lane in flight; a count of 0 / 1 flags: exact in any order) --
            // the loop below is four dependent round trips at the recurrent update's 1 920 rows, in every one of its 1 920 workgroups
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = p.filled[min(lane + 64 * u, p.B - 1)];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += lane + 64 * u < p.B ? v[u] : 0.f;
            i = p.B;
        }
        for (; i + 64 * 7 < p.B; i += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p.filled[i + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < p.B; i += 64) s += p.filled[i];
        inv_norm = 1.f / wave_sum(s);
        fl = p.filled[b];
    }
    // ---- per-agent part (lanes < N), same operations as qmix_kernel
    float qe = 0.f, qn = 0.f, mask = 0.f;
    int a_taken = 0;
    if (lane < N) {
        mask = l_mask * fl;
        a_taken = (int)l_act;
        float qa = qev[0];
#pragma unroll
        for (int j = 1; j < QM_MAXA; ++j) qa = (j == a_taken) ? qev[j] : qa;
        qe = qa * mask;
        const bool use_av = p.avail_next != nullptr;
        if (p.double_q) {
            int best = 0; float bv = (use_av && avv[0] == 0.f) ? -1e10f : qsv[0];
#pragma unroll
            for (int j = 1; j < QM_MAXA; ++j) {
                if (j < A) {
                    const float v = (use_av && avv[j] == 0.f) ? -1e10f : qsv[j];
                    if (v > bv) { bv = v; best = j; }
                }
            }
            float qb = qtv[0], ab = avv[0];
#pragma unroll
            for (int j = 1; j < QM_MAXA; ++j) { qb = (j == best) ? qtv[j] : qb; ab = (j == best) ? avv[j] : ab; }
            qn = (use_av && ab == 0.f) ? -1e10f : qb;
        } else {
            qn = (use_av && avv[0] == 0.f) ? -1e10f : qtv[0];
#pragma unroll
            for (int j = 1; j < QM_MAXA; ++j)
                if (j < A) qn = fmaxf(qn, (use_av && avv[j] == 0.f) ? -1e10f : qtv[j]);
        }
        qn *= mask;
    }
    // ---- mixing networks (q_mix_head.py:78-95)
    float pre_e = 0.f, pre_t = 0.f, w2e = 0.f, w2t = 0.f;
    if (lane < H) { pre_e = l_eb1; pre_t = l_tb1; }
#pragma unroll
    for (int n = 0; n < QM_MAXN; ++n) {
        if (n < N) {
            const float qen = __shfl(qe, n, 64), qnn = __shfl(qn, n, 64);
            if (lane < H) {
                pre_e += qen * fabsf(er[n]);
                pre_t += qnn * fabsf(tr[n]);
            }
        }
    }
    float hid_e = 0.f, hid_t = 0.f;
    if (lane < H) {
        hid_e = elu_f(pre_e); hid_t = elu_f(pre_t);
        w2e = fabsf(l_ew2); w2t = fabsf(l_tw2);
    }
    const float q_tot_e = wave_sum(hid_e * w2e) + l_eb2;
    const float q_tot_n = wave_sum(hid_t * w2t) + l_tb2;
    float r = 0.f, dn = 1.f;
    if (lane < N) { r = l_rew; dn = l_term != 0.f ? 1.f : 0.f; }
    const float r_tot = wave_sum(r) / (float)N;
    float all_d = dn;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) all_d = fminf(all_d, __shfl_xor(all_d, off, 64));
    const float y = r_tot + (1.f - all_d) * p.gamma * q_tot_n;
    const float td = (q_tot_e - y) * fl;
    const float dq_tot = p.filled ? 2.f * td * fl * inv_norm : 2.f * td / (float)p.B;
    float* d_raw = p.d_e_raw + (size_t)b * p.ld_e2;
    float d_pre = 0.f;
    if (lane < H) {
        const float sgn2 = (l_ew2 > 0.f) - (l_ew2 < 0.f);
        d_raw[N * H + lane] = dq_tot * hid_e * sgn2;
        d_pre = dq_tot * w2e * (pre_e > 0.f ? 1.f : expf(pre_e));
        p.d_e_b1[(size_t)b * p.ld_e1 + lane] = d_pre;
    }
    if (lane == 0) d_raw[N * H + H] = dq_tot;
#pragma unroll
    for (int n = 0; n < QM_MAXN; ++n) {
        if (n < N) {
            const float qen = __shfl(qe, n, 64);
            float contrib = 0.f;
            if (lane < H) {
                const float sgn1 = (er[n] > 0.f) - (er[n] < 0.f);
                d_raw[n * H + lane] = qen * d_pre * sgn1;
                contrib = d_pre * fabsf(er[n]);
            }
            const float dqe = wave_sum(contrib);
            if (lane == n) {
                float* dq = p.d_q + ((size_t)b * N + n) * p.ldq;
                for (int j = 0; j < A; ++j) dq[j] = (j == a_taken) ? dqe * mask : 0.f;
            }
        }
    }
    if (lane == 0) {
        double* q = p.partials + (size_t)b * 8;
        q[0] = (double)td * td; q[1] = q_tot_e; q[2] = p.filled ? (double)fl : 0.0;
        for (int j = 3; j < 8; ++j) q[j] = 0.0;
        if (p.diag) { p.diag[b] = q_tot_e; p.diag[p.B + b] = q_tot_n; p.diag[2 * (size_t)p.B + b] = y; }
    }
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_dqn_td(const xrl_dqn_td_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->q_eval && p->q_next && p->actions && p->rewards && p->terminals && p->d_q && p->partials);
    XRL_CHECK_ARG(p->M > 0 && p->A > 0 && p->ld >= p->A + (p->dueling ? 1 : 0) && p->n_split >= 1);
    hipLaunchKernelGGL(dqn_td_kernel, dim3(p->n_split), dim3(256), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_dqn_head_td(const xrl_dqn_head_td_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->h_eval && p->h_target && p->w_eval && p->b_eval && p->w_target && p->b_target && p->actions && p->rewards &&
                  p->terminals && p->q_eval && p->q_target && p->d_q && p->d_h && p->partials);
    XRL_CHECK_ARG(p->M > 0 && p->A > 0 && p->A <= 64 && p->H > 0 && p->ld_h >= p->H && p->ld_q >= p->A);
    hipLaunchKernelGGL(dqn_head_td_kernel, dim3(p->M < 1024 ? p->M : 1024), dim3(256), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_dqn_tail_td(const xrl_dqn_tail_td_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->y_eval && p->y_target && p->feat_eval && p->w1_eval && p->b1_eval && p->w1_target && p->b1_target &&
                  p->w2_eval && p->b2_eval && p->w2_target && p->b2_target && p->actions && p->rewards && p->terminals &&
                  p->q_eval && p->q_target && p->d_q && p->h_eval && p->d_h && p->dy && p->partials);
    XRL_CHECK_ARG(p->M > 0 && p->M <= 65535 && p->A > 0 && p->A <= 64 && p->F == TAIL_F && p->H >= 1 && p->H <= TAIL_HMAX);
    XRL_CHECK_ARG(p->P > 0 && p->P <= TAIL_W * TAIL_PQ && p->ld_h >= p->H && p->ld_q >= p->A && p->ld_f >= p->F);
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p->w1_eval) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->w1_target) & 15) == 0);
    XRL_CHECK_ARG(p->slabs == nullptr || ((reinterpret_cast<uintptr_t>(p->slabs) & 15) == 0 && p->slab_stride % 4 == 0 && p->off_w1 % 4 == 0 &&
                                          p->off_w1 >= 0 && p->off_b1 >= 0 && p->off_w2 >= 0 && p->off_b2 >= 0));
    hipLaunchKernelGGL(dqn_tail_td_kernel, dim3(p->M), dim3(TAIL_T), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_dqn_act_tail(const xrl_dqn_act_tail_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->y && p->w1 && p->b1 && p->w2 && p->b2 && p->action);
    XRL_CHECK_ARG(p->n > 0 && p->n <= 65535 && p->A > 0 && p->A <= 64 && p->F == TAIL_F && p->H >= 1 && p->H <= TAIL_HMAX);
    XRL_CHECK_ARG(p->P > 0 && p->P <= TAIL_W * TAIL_PQ && (!p->q || p->ld_q >= p->A) && (!p->feat || p->ld_f >= p->F));
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p->w1) & 15) == 0);
    hipLaunchKernelGGL(dqn_act_tail_kernel, dim3(p->n), dim3(TAIL_T), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_qmix_mix_td(const xrl_qmix_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->q_eval && p->q_next && p->actions && p->agent_mask && p->rewards && p->terminals);
    XRL_CHECK_ARG(p->d_q && p->partials && p->mixer >= 0 && p->mixer <= 2);
    XRL_CHECK_ARG(p->B > 0 && p->N > 0 && p->N <= 64 && p->A > 0 && p->ldq >= p->A);
    XRL_CHECK_ARG(!p->double_q || p->q_next_eval);
    if (p->mixer == 0) {
        XRL_CHECK_ARG(p->e_b1 && p->e_raw && p->t_b1 && p->t_raw && p->d_e_b1 && p->d_e_raw && p->H > 0 && p->H <= 64);
        XRL_CHECK_ARG(p->ld_e2 >= p->N * p->H + p->H + 1 && p->ld_t2 >= p->N * p->H + p->H + 1);
    }
    XRL_CHECK_ARG(p->mixer != 2 || !p->filled);
    if (p->mixer == 0 && p->N <= QM_MAXN && p->A <= QM_MAXA)
        hipLaunchKernelGGL(qmix_prefetch_kernel, dim3(p->B), dim3(64), 0, as_stream(stream), *p);
    else
        hipLaunchKernelGGL(qmix_kernel, dim3(p->B), dim3(64), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_sync_target(const float* params, float* target, int64_t P, const xrl_adam_state_t* state,
                               int sync_frequency, xrl_stream_t stream) {
    XRL_CHECK_ARG(params && target && state && P > 0 && sync_frequency > 0);
    int nb = (int)((P + 255) / 256);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sync_target_kernel, dim3(nb), dim3(256), 0, as_stream(stream), params, target, P, state,
                       sync_frequency);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
