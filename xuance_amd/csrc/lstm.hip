// One-layer LSTM over whole sequences (forward + back-propagation through time), time-major: the `rnn: "LSTM"` option of
// Basic_RNN (xuance/torch/rl_models/representations/rnn.py:45-47,59-66; lstm_block, rl_models/modules/layers.py:101-125).
// Cell arithmetic is torch.nn.LSTM's, gate order i | f | g | o:
//   i = sigmoid(.)  f = sigmoid(.)  g = tanh(.)  o = sigmoid(.)   of   gi + W_hh h + b_hh;   c' = f c + i g;   h' = o tanh(c')
// Same mapping as csrc/gru.hip (see there): the input-side products and every weight gradient are chip-wide GEMMs; the
// recurrence runs one two-wave workgroup per sequence, lane j = hidden unit j, each wave holding half of unit j's four
// W_hh rows (forward) / half of column j (backward) in 128 registers, packed fp32 FMAs, operands that do not depend on
// the recurrence prefetched two steps ahead.  The gate pre-activation gradient is the same for the input and the hidden
// side, so the backward pass writes one d_gates buffer (dW_ih = d_gates^T x, dW_hh = d_gates^T h_prev).
#include "common.h"

namespace xrl {

constexpr int LH = 64;

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f lpk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float lsigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float ltanh_f(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
__device__ __forceinline__ void lwave_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__global__ void __launch_bounds__(128) lstm_forward_kernel(xrl_lstm_fwd_t p) {
    __shared__ __attribute__((aligned(16))) float hl[2][LH / 2];    // per wave: its k-half of h as pairs (h[k], h[k+16])
    __shared__ float xch[2][2][4][LH];                              // [step parity][wave][gate][unit] partial sums
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = p.R, T1 = p.T1;
    const bool second = (int)blockIdx.x >= R;                       // second problem of the launch (target network)
    const int row = second ? blockIdx.x - R : blockIdx.x;
    const float* w_hh = second ? p.w_hh2 : p.w_hh;
    const float* b_hh = second ? p.b_hh2 : p.b_hh;
    const float* gi_base = second ? p.gi2 : p.gi;
    float* hs = second ? p.hs2 : p.hs;
    float* cs = second ? nullptr : p.cs;
    float* gates = (second || w) ? nullptr : p.gates;
    v2f wg[4][LH / 4];                               // gate x (W[g*H + j][k], W[g*H + j][k+16]), k in this wave's half
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4* a = reinterpret_cast<const float4*>(w_hh + (size_t)(g * LH + j) * LH + 32 * w);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 lo = a[q], hi = a[q + 4];
            wg[g][4 * q] = {lo.x, hi.x}; wg[g][4 * q + 1] = {lo.y, hi.y}; wg[g][4 * q + 2] = {lo.z, hi.z}; wg[g][4 * q + 3] = {lo.w, hi.w};
        }
    }
    float bg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bg[g] = b_hh[g * LH + j];
    float h = (p.h0 && !second) ? p.h0[(size_t)row * LH + j] : 0.f;
    float c = (p.c0 && !second) ? p.c0[(size_t)row * LH + j] : 0.f;
    if (p.reset && !second && p.reset[row] != 0.f) { h = 0.f; c = 0.f; }    // init_rnn_states_item (rnn.py:86-92)
    if (w == 0) { hs[(size_t)row * LH + j] = h; if (cs) cs[(size_t)row * LH + j] = c; }
    const bool mine = (j >> 5) == w;
    const int pos = ((j & 15) << 1) | ((j >> 4) & 1);
    const float* gi = gi_base + (size_t)row * p.ld_gi + j;
    const size_t gstep = (size_t)R * p.ld_gi;
    float gq[3][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* g2 = gi + (size_t)min(s, T1 - 1) * gstep;
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[s][g] = g2[g * LH];
    }
    int t = 0;
#define LSTM_FWD_STEP(CUR, NXT)                                                                                 \
    {                                                                                                           \
        {                                                                                                       \
            const float* g2 = gi + (size_t)min(t + 2, T1 - 1) * gstep;                                          \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) gq[NXT][g] = g2[g * LH];                              \
        }                                                                                                       \
        if (mine) hl[w][pos] = h;                                                                               \
        lwave_fence();                                                                                          \
        v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                          \
        float4 hq[LH / 8];                                                                                      \
        _Pragma("unroll") for (int q = 0; q < LH / 8; ++q) hq[q] = reinterpret_cast<const float4*>(hl[w])[q];   \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int q = 0; q < LH / 8; ++q) {                                                    \
            const v2f h0 = {hq[q].x, hq[q].y}, h1 = {hq[q].z, hq[q].w};                                         \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                     \
                acc[g] = lpk_fma(wg[g][2 * q], h0, acc[g]); acc[g] = lpk_fma(wg[g][2 * q + 1], h1, acc[g]);     \
            }                                                                                                   \
        }                                                                                                       \
        float(*x)[4][LH] = xch[t & 1];                                                                          \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) x[w][g][j] = acc[g].x + acc[g].y;                         \
        lds_barrier();                                                                                          \
        float pre[4];                                                                                           \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) pre[g] = gq[CUR][g] + (bg[g] + (x[0][g][j] + x[1][g][j])); \
        const float ig = lsigmoid_f(pre[0]), fg = lsigmoid_f(pre[1]), gg = ltanh_f(pre[2]), og = lsigmoid_f(pre[3]); \
        c = fg * c + ig * gg;                                                                                   \
        h = og * ltanh_f(c);                                                                                    \
        if (w == 0) {                                                                                           \
            const size_t o1 = ((size_t)(t + 1) * R + row) * LH + j;                                             \
            hs[o1] = h;                                                                                         \
            if (cs) cs[o1] = c;                                                                                 \
        }                                                                                                       \
        if (gates) {                                                                                            \
            float* gp = gates + ((size_t)t * R + row) * 4 * LH;                                                 \
            gp[j] = ig; gp[LH + j] = fg; gp[2 * LH + j] = gg; gp[3 * LH + j] = og;                              \
        }                                                                                                       \
        if (++t >= T1) break;                                                                                   \
    }
    for (;;) {
        LSTM_FWD_STEP(0, 2)
        LSTM_FWD_STEP(1, 0)
        LSTM_FWD_STEP(2, 1)
    }
#undef LSTM_FWD_STEP
    if (!second && w == 0) {
        if (p.h_last) p.h_last[(size_t)row * LH + j] = h;
        if (p.c_last) p.c_last[(size_t)row * LH + j] = c;
    }
}

// BPTT.  Lane k owns hidden unit k; wave w holds W_hh[e][k] for its 128 of the 256 gate rows as pairs (e, e + 64).
__global__ void __launch_bounds__(128) lstm_backward_kernel(xrl_lstm_bwd_t p) {
    __shared__ __attribute__((aligned(16))) float gl[2][2 * LH];    // per wave: its 128 gate gradients, pairs (g[e], g[e+64])
    __shared__ float xch[2][2][LH];
    const int row = blockIdx.x, k = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = p.R, T1 = p.T1;
    v2f wc[LH];                                       // rows 128 w + e and 128 w + e + 64, e < 64
#pragma unroll
    for (int e = 0; e < LH; ++e)
        wc[e] = {p.w_hh[(size_t)(128 * w + e) * LH + k], p.w_hh[(size_t)(128 * w + e + 64) * LH + k]};
    // d_gates = di | df | dg | do (64 each): wave 0 owns (di, df) = entries e and e + 64 of its half, wave 1 (dg, do)
    float dh_carry = 0.f, dc_carry = 0.f;
    float op[3][7];                                   // i f g o | c_{t-1} | c_t | d_hs
    const float* gbase = p.gates + (size_t)row * 4 * LH + k;
    const float* cbase = p.cs + (size_t)row * LH + k;                         // slot t = c_{t-1}
    const float* dbase = p.d_hs + (size_t)row * p.ld_dhs + k;
#define LSTM_BWD_LOAD(S, TT)                                                                                    \
    {                                                                                                           \
        const size_t tt = (size_t)max((TT), 0);                                                                 \
        const float* g = gbase + tt * R * 4 * LH;                                                               \
        op[S][0] = g[0]; op[S][1] = g[LH]; op[S][2] = g[2 * LH]; op[S][3] = g[3 * LH];                          \
        op[S][4] = cbase[tt * R * LH]; op[S][5] = cbase[(tt + 1) * R * LH]; op[S][6] = dbase[tt * R * p.ld_dhs]; \
    }
    int t = T1 - 1;
    LSTM_BWD_LOAD(0, t)
    LSTM_BWD_LOAD(1, t - 1)
#define LSTM_BWD_STEP(CUR, NXT)                                                                                 \
    {                                                                                                           \
        LSTM_BWD_LOAD(NXT, t - 2)                                                                               \
        const float ig = op[CUR][0], fg = op[CUR][1], gg = op[CUR][2], og = op[CUR][3], cp = op[CUR][4];        \
        const float tc = ltanh_f(op[CUR][5]);                                                                   \
        const float dh = op[CUR][6] + dh_carry;                                                                 \
        const float dc = dh * og * (1.f - tc * tc) + dc_carry;                                                  \
        const float di = dc * gg * ig * (1.f - ig), df = dc * cp * fg * (1.f - fg);                             \
        const float dg = dc * ig * (1.f - gg * gg), dq = dh * tc * og * (1.f - og);                             \
        if (w == 0) {                                                                                           \
            float* d = p.d_gates + ((size_t)t * R + row) * p.ld_dg;                                             \
            d[k] = di; d[LH + k] = df; d[2 * LH + k] = dg; d[3 * LH + k] = dq;                                  \
        }                                                                                                       \
        gl[w][2 * k] = (w == 0) ? di : dg;                                                                      \
        gl[w][2 * k + 1] = (w == 0) ? df : dq;                                                                  \
        lwave_fence();                                                                                          \
        v2f acc = {0.f, 0.f};                                                                                   \
        float4 g4[LH / 2];                                                                                      \
        _Pragma("unroll") for (int q = 0; q < LH / 2; ++q) g4[q] = reinterpret_cast<const float4*>(gl[w])[q];   \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int q = 0; q < LH / 2; ++q) {                                                    \
            acc = lpk_fma(v2f{g4[q].x, g4[q].y}, wc[2 * q], acc); acc = lpk_fma(v2f{g4[q].z, g4[q].w}, wc[2 * q + 1], acc); \
        }                                                                                                       \
        float(*x)[LH] = xch[t & 1];                                                                             \
        x[w][k] = acc.x + acc.y;                                                                                \
        lds_barrier();                                                                                          \
        dh_carry = x[0][k] + x[1][k];                                                                           \
        dc_carry = dc * fg;                                                                                     \
        if (--t < 0) break;                                                                                     \
    }
    for (;;) {
        LSTM_BWD_STEP(0, 2)
        LSTM_BWD_STEP(1, 0)
        LSTM_BWD_STEP(2, 1)
    }
#undef LSTM_BWD_STEP
#undef LSTM_BWD_LOAD
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_lstm_forward(const xrl_lstm_fwd_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->gi && p->w_hh && p->b_hh && p->hs);
    XRL_CHECK_ARG(p->H == LH && p->R > 0 && p->T1 > 0 && p->ld_gi >= 4 * LH);
    XRL_CHECK_ARG(!p->gates || p->cs);                                     // what BPTT needs comes as a pair
    const bool dual = p->gi2 != nullptr;
    XRL_CHECK_ARG(!dual || (p->w_hh2 && p->b_hh2 && p->hs2));
    hipLaunchKernelGGL(lstm_forward_kernel, dim3(dual ? 2 * p->R : p->R), dim3(128), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_lstm_backward(const xrl_lstm_bwd_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->d_hs && p->cs && p->gates && p->w_hh && p->d_gates);
    XRL_CHECK_ARG(p->H == LH && p->R > 0 && p->T1 > 0 && p->ld_dhs >= LH && p->ld_dg >= 4 * LH);
    hipLaunchKernelGGL(lstm_backward_kernel, dim3(p->R), dim3(128), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
