// One-layer GRU over whole sequences (forward + back-propagation through time), time-major, for the recurrent agents of
// QMIX: Basic_RNN (xuance/torch/rl_models/representations/rnn.py:52-77) = mlp -> nn.GRU(batch_first) built by
// gru_block (rl_models/modules/layers.py:79-98).  Cell arithmetic is torch.nn.GRU's:
//   r = sigmoid(gi_r + W_hr h + b_hr)   z = sigmoid(gi_z + W_hz h + b_hz)
//   n = tanh(gi_n + r * (W_hn h + b_hn))   h' = (h - n) * z + n            with gi = W_i x + b_i.
//
// Mapping: the input-side products gi (all steps at once) and every weight gradient are plain GEMMs over T1*R rows and
// run through xrl_linear_* on the matrix cores.  What is left is the serial part: per step a [R, 64] x [64, 192] product
// whose R rows (sequences) are independent.  At the sizes of this path (R = batch 32 x 3 agents = 96 sequences, 61
// steps) that chain is latency-bound, not throughput-bound, so the recurrence runs ONE WAVEFRONT PER SEQUENCE with no
// workgroup barrier and no cross-wave traffic at all: lane j owns hidden unit j and keeps its three rows of W_hh (forward)
// or its column of W_hh (backward) in 192 registers for the whole sequence; the previous hidden state (forward) or the
// gate gradients (backward) are broadcast to the wave through 256 / 768 bytes of LDS.  A step costs 96 v_pk_fma_f32 per
// lane + 16 (48) broadcast ds_read_b128; a 32-row MFMA tile would need 6 output tiles x 32 chained v_mfma_f32_32x32x2 per
// step on one CU and leave all but 3 CUs idle.  Per-step operands that do not depend on the recurrence (gi; saved gates
// and d_hs in the backward pass) are loaded two steps ahead into a statically indexed 3-slot register ring, so no step
// waits on a global round trip.
#include "common.h"

namespace xrl {

constexpr int GH = 64;    // hidden width == wavefront width

typedef float v2f __attribute__((ext_vector_type(2)));
// two fp32 FMAs per lane and issue slot (v_pk_fma_f32): the dot products run as two half-sums (k < 32 | k >= 32)
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the branchy libm expf / tanhf / IEEE division: the gate non-linearities sit
// on the serial chain of every step.  Absolute error ~1e-7, far inside the 1e-5 parity tolerance.
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Two waves per sequence split every dot product of a step in halves (forward: k < 32 | k >= 32; backward: the first /
// second 96 of the 192 gate gradients): 96 instead of 192 weight registers per lane leave room to have a whole step's
// broadcast LDS reads in flight at once (with all 192 weights in one wave only ~2 ds_read_b128 fit, and their latency was
// exposed 24 times per step).  Everything else of a step is computed redundantly by both waves, so the only cross-wave
// traffic is the partial sums: one LDS exchange + one s_barrier per step (double-buffered by step parity).
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__global__ void __launch_bounds__(128) gru_forward_kernel(xrl_gru_fwd_t p) {
    __shared__ __attribute__((aligned(16))) float hl[2][GH / 2];    // per wave: its k-half of h as pairs (h[k], h[k+16])
    __shared__ float xch[2][2][3][GH];                              // [step parity][wave][gate][unit] partial sums
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = p.R, T1 = p.T1;
    // second problem of the launch (same shapes, other weights / inputs: the target network of a QMIX update)
    const bool second = (int)blockIdx.x >= R;
    const int row = second ? blockIdx.x - R : blockIdx.x;
    const float* w_hh = second ? p.w_hh2 : p.w_hh;
    const float* b_hh = second ? p.b_hh2 : p.b_hh;
    const float* gi_base = second ? p.gi2 : p.gi;
    float* hs = second ? p.hs2 : p.hs;
    float* gates = (second || w) ? nullptr : p.gates;               // wave 0 does the global stores
    v2f wr[GH / 4], wz[GH / 4], wn[GH / 4];          // (W[j][k], W[j][k+16]), k in this wave's half, of unit j's three rows
    {
        const float4* a = reinterpret_cast<const float4*>(w_hh + (size_t)j * GH + 32 * w);
        const float4* b = reinterpret_cast<const float4*>(w_hh + (size_t)(GH + j) * GH + 32 * w);
        const float4* c = reinterpret_cast<const float4*>(w_hh + (size_t)(2 * GH + j) * GH + 32 * w);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xl = a[q], xh = a[q + 4], yl = b[q], yh = b[q + 4], zl = c[q], zh = c[q + 4];
            wr[4 * q] = {xl.x, xh.x}; wr[4 * q + 1] = {xl.y, xh.y}; wr[4 * q + 2] = {xl.z, xh.z}; wr[4 * q + 3] = {xl.w, xh.w};
            wz[4 * q] = {yl.x, yh.x}; wz[4 * q + 1] = {yl.y, yh.y}; wz[4 * q + 2] = {yl.z, yh.z}; wz[4 * q + 3] = {yl.w, yh.w};
            wn[4 * q] = {zl.x, zh.x}; wn[4 * q + 1] = {zl.y, zh.y}; wn[4 * q + 2] = {zl.z, zh.z}; wn[4 * q + 3] = {zl.w, zh.w};
        }
    }
    const float br = b_hh[j], bz = b_hh[GH + j], bn = b_hh[2 * GH + j];
    float h = (p.h0 && !second) ? p.h0[(size_t)row * GH + j] : 0.f;
    if (p.reset && !second && p.reset[row] != 0.f) h = 0.f;                 // init_rnn_states_item (rnn.py:86-92)
    if (w == 0) hs[(size_t)row * GH + j] = h;                               // slot 0
    const bool mine = (j >> 5) == w;                                        // this lane's unit lies in this wave's k-half
    const int pos = ((j & 15) << 1) | ((j >> 4) & 1);                       // k-in-half kk: pair (kk, kk+16) -> 2*(kk&15) + (kk>>4)
    const float* gi = gi_base + (size_t)row * p.ld_gi + j;
    const size_t gstep = (size_t)R * p.ld_gi;
    // input-side gates of step t are loaded two steps ahead into a 3-slot register ring (they do not depend on h)
    float gq[3][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* g2 = gi + (size_t)min(s, T1 - 1) * gstep;
        gq[s][0] = g2[0]; gq[s][1] = g2[GH]; gq[s][2] = g2[2 * GH];
    }
    int t = 0;
#define GRU_FWD_STEP(CUR, NXT)                                                                                  \
    {                                                                                                           \
        {                                                                                                       \
            const float* g2 = gi + (size_t)min(t + 2, T1 - 1) * gstep;                                          \
            gq[NXT][0] = g2[0]; gq[NXT][1] = g2[GH]; gq[NXT][2] = g2[2 * GH];                                   \
        }                                                                                                       \
        if (mine) hl[w][pos] = h;                                                                               \
        wave_lds_fence();                                                                                       \
        v2f ar = {0.f, 0.f}, az = {0.f, 0.f}, an = {0.f, 0.f};                                                  \
        float4 hq[GH / 8];               /* all broadcast reads of the step in flight before the first FMA */   \
        _Pragma("unroll") for (int q = 0; q < GH / 8; ++q) hq[q] = reinterpret_cast<const float4*>(hl[w])[q];   \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int q = 0; q < GH / 8; ++q) {                                                    \
            const float4 hv = hq[q];                                                                            \
            const v2f h0 = {hv.x, hv.y}, h1 = {hv.z, hv.w};                                                     \
            ar = pk_fma(wr[2 * q], h0, ar); ar = pk_fma(wr[2 * q + 1], h1, ar);                                 \
            az = pk_fma(wz[2 * q], h0, az); az = pk_fma(wz[2 * q + 1], h1, az);                                 \
            an = pk_fma(wn[2 * q], h0, an); an = pk_fma(wn[2 * q + 1], h1, an);                                 \
        }                                                                                                       \
        float(*x)[3][GH] = xch[t & 1];                                                                          \
        x[w][0][j] = ar.x + ar.y; x[w][1][j] = az.x + az.y; x[w][2][j] = an.x + an.y;                           \
        lds_barrier();                                                                                          \
        const float sr = br + (x[0][0][j] + x[1][0][j]), sz = bz + (x[0][1][j] + x[1][1][j]);                   \
        const float hn = bn + (x[0][2][j] + x[1][2][j]);                                                        \
        const float r = sigmoid_f(gq[CUR][0] + sr);                                                             \
        const float z = sigmoid_f(gq[CUR][1] + sz);                                                             \
        const float n = tanh_f(gq[CUR][2] + r * hn);                                                            \
        h = (h - n) * z + n;                                                                                    \
        if (w == 0) hs[((size_t)(t + 1) * R + row) * GH + j] = h;                                               \
        if (gates) {                                                                                            \
            float* g = gates + ((size_t)t * R + row) * 4 * GH;                                                  \
            g[j] = r; g[GH + j] = z; g[2 * GH + j] = n; g[3 * GH + j] = hn;                                     \
        }                                                                                                       \
        if (++t >= T1) break;                                                                                   \
    }
    for (;;) {
        GRU_FWD_STEP(0, 2)
        GRU_FWD_STEP(1, 0)
        GRU_FWD_STEP(2, 1)
    }
#undef GRU_FWD_STEP
    if (p.h_last && !second && w == 0) p.h_last[(size_t)row * GH + j] = h;
}

// BPTT.  Lane k owns hidden unit k: its wave's half of column k of W_hh (96 of 192 values), as pairs (W[e][k], W[e+48][k]).
__global__ void __launch_bounds__(128) gru_backward_kernel(xrl_gru_bwd_t p) {
    __shared__ __attribute__((aligned(16))) float gl[2][3 * GH / 2];   // per wave: its 96 gate gradients, pairs (g[e], g[e+48])
    __shared__ float xch[2][2][GH];                                    // [step parity][wave][unit] partial sums
    const int row = blockIdx.x, k = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = p.R, T1 = p.T1;
    v2f wc[3 * GH / 4];
#pragma unroll
    for (int e = 0; e < 3 * GH / 4; ++e)
        wc[e] = {p.w_hh[(size_t)(96 * w + e) * GH + k], p.w_hh[(size_t)(96 * w + e + 48) * GH + k]};
    // the 192-vector is dr_pre[0..63] | dz_pre[0..63] | dhn[0..63]; wave 0 needs entries 0..95, wave 1 entries 96..191.
    // local entry le (0..95) lives at 2*le (le < 48) or 2*(le-48)+1.
    auto lpos = [](int le) { return le < 48 ? 2 * le : 2 * (le - 48) + 1; };
    const int pa = lpos(w == 0 ? k : 32 + k);                         // wave 0: dr_pre[k] -> le k;  wave 1: dhn[k] -> le 32+k
    const bool has_b = (w == 0) ? (k < 32) : (k >= 32);               // dz_pre[k]: wave 0 takes k < 32 (le 64+k), wave 1 k >= 32 (le k-32)
    const int pb = lpos(w == 0 ? 64 + (k & 31) : (k & 31));
    float carry = 0.f;
    // operands of step t: r z n hn | h_{t-1} | d_hs, loaded two steps ahead (they do not depend on the carry)
    float op[3][6];
    const float* gbase = p.gates + (size_t)row * 4 * GH + k;
    const float* hbase = p.hs + (size_t)row * GH + k;                        // slot t = h_{t-1}
    const float* dbase = p.d_hs + (size_t)row * p.ld_dhs + k;
#define GRU_BWD_LOAD(S, TT)                                                                                     \
    {                                                                                                           \
        const size_t tt = (size_t)max((TT), 0);                                                                 \
        const float* g = gbase + tt * R * 4 * GH;                                                               \
        op[S][0] = g[0]; op[S][1] = g[GH]; op[S][2] = g[2 * GH]; op[S][3] = g[3 * GH];                          \
        op[S][4] = hbase[tt * R * GH]; op[S][5] = dbase[tt * R * p.ld_dhs];                                     \
    }
    int t = T1 - 1;
    GRU_BWD_LOAD(0, t)
    GRU_BWD_LOAD(1, t - 1)
#define GRU_BWD_STEP(CUR, NXT)                                                                                  \
    {                                                                                                           \
        GRU_BWD_LOAD(NXT, t - 2)                                                                                \
        const float r = op[CUR][0], z = op[CUR][1], n = op[CUR][2], hn = op[CUR][3], hp = op[CUR][4];           \
        const float dh = op[CUR][5] + carry;                                                                    \
        const float dn_pre = dh * (1.f - z) * (1.f - n * n);                                                    \
        const float dz_pre = dh * (hp - n) * z * (1.f - z);                                                     \
        const float dr_pre = dn_pre * hn * r * (1.f - r);                                                       \
        const float dhn = dn_pre * r;                                                                           \
        if (w == 0) {                                                                                           \
            const size_t o = (size_t)t * R + row;                                                               \
            float* dgi = p.d_gi + o * p.ld_dgi;                                                                 \
            dgi[k] = dr_pre; dgi[GH + k] = dz_pre; dgi[2 * GH + k] = dn_pre;                                    \
            float* dgh = p.d_gh + o * 3 * GH;                                                                   \
            dgh[k] = dr_pre; dgh[GH + k] = dz_pre; dgh[2 * GH + k] = dhn;                                       \
        }                                                                                                       \
        gl[w][pa] = (w == 0) ? dr_pre : dhn;                                                                    \
        if (has_b) gl[w][pb] = dz_pre;                                                                          \
        wave_lds_fence();                                                                                       \
        v2f acc = {0.f, 0.f};                                                                                   \
        float4 gq4[3 * GH / 8];          /* all 24 broadcast reads in flight before the first FMA */            \
        _Pragma("unroll") for (int q = 0; q < 3 * GH / 8; ++q) gq4[q] = reinterpret_cast<const float4*>(gl[w])[q]; \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int q = 0; q < 3 * GH / 8; ++q) {                                                \
            const float4 gv = gq4[q];                                                                           \
            acc = pk_fma(v2f{gv.x, gv.y}, wc[2 * q], acc); acc = pk_fma(v2f{gv.z, gv.w}, wc[2 * q + 1], acc);   \
        }                                                                                                       \
        float(*x)[GH] = xch[t & 1];                                                                             \
        x[w][k] = acc.x + acc.y;                                                                                \
        lds_barrier();                                                                                          \
        carry = dh * z + (x[0][k] + x[1][k]);                                                                   \
        if (--t < 0) break;                                                                                     \
    }
    for (;;) {
        GRU_BWD_STEP(0, 2)
        GRU_BWD_STEP(1, 0)
        GRU_BWD_STEP(2, 1)
    }
#undef GRU_BWD_STEP
#undef GRU_BWD_LOAD
    if (p.d_h0 && w == 0) p.d_h0[(size_t)row * GH + k] = carry;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_gru_forward(const xrl_gru_fwd_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->gi && p->w_hh && p->b_hh && p->hs);
    XRL_CHECK_ARG(p->H == GH);                       // one lane per hidden unit (3m.yaml: recurrent_hidden_size 64)
    XRL_CHECK_ARG(p->R > 0 && p->T1 > 0 && p->ld_gi >= 3 * GH);
    const bool dual = p->gi2 != nullptr;
    XRL_CHECK_ARG(!dual || (p->w_hh2 && p->b_hh2 && p->hs2));
    hipLaunchKernelGGL(gru_forward_kernel, dim3(dual ? 2 * p->R : p->R), dim3(128), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_gru_backward(const xrl_gru_bwd_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->d_hs && p->hs && p->gates && p->w_hh && p->d_gi && p->d_gh);
    XRL_CHECK_ARG(p->H == GH);
    XRL_CHECK_ARG(p->R > 0 && p->T1 > 0 && p->ld_dhs >= GH && p->ld_dgi >= 3 * GH);
    hipLaunchKernelGGL(gru_backward_kernel, dim3(p->R), dim3(128), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
