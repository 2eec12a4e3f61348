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
// gate gradients (backward) are broadcast to the wave through 256 / 768 bytes of LDS.  A step costs 192 FMAs per lane +
// 16 (48) broadcast ds_read_b128; a 32-row MFMA tile would need 6 output tiles x 32 chained v_mfma_f32_32x32x2 per step
// on one CU and leave all but 3 CUs idle.
#include "common.h"

namespace xrl {

constexpr int GH = 64;    // hidden width == wavefront width

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(64) gru_forward_kernel(xrl_gru_fwd_t p) {
    __shared__ __attribute__((aligned(16))) float hl[GH];
    const int j = threadIdx.x;
    const int R = p.R;
    // second problem of the launch (same shapes, other weights / inputs: the target network of a QMIX update)
    const bool second = (int)blockIdx.x >= R;
    const int row = second ? blockIdx.x - R : blockIdx.x;
    const float* w_hh = second ? p.w_hh2 : p.w_hh;
    const float* b_hh = second ? p.b_hh2 : p.b_hh;
    const float* gi_base = second ? p.gi2 : p.gi;
    float* hs = second ? p.hs2 : p.hs;
    float* gates = second ? nullptr : p.gates;
    float wr[GH], wz[GH], wn[GH];
    {
        const float4* a = reinterpret_cast<const float4*>(w_hh + (size_t)j * GH);
        const float4* b = reinterpret_cast<const float4*>(w_hh + (size_t)(GH + j) * GH);
        const float4* c = reinterpret_cast<const float4*>(w_hh + (size_t)(2 * GH + j) * GH);
#pragma unroll
        for (int q = 0; q < GH / 4; ++q) {
            const float4 x = a[q], y = b[q], z = c[q];
            wr[4 * q] = x.x; wr[4 * q + 1] = x.y; wr[4 * q + 2] = x.z; wr[4 * q + 3] = x.w;
            wz[4 * q] = y.x; wz[4 * q + 1] = y.y; wz[4 * q + 2] = y.z; wz[4 * q + 3] = y.w;
            wn[4 * q] = z.x; wn[4 * q + 1] = z.y; wn[4 * q + 2] = z.z; wn[4 * q + 3] = z.w;
        }
    }
    const float br = b_hh[j], bz = b_hh[GH + j], bn = b_hh[2 * GH + j];
    float h = (p.h0 && !second) ? p.h0[(size_t)row * GH + j] : 0.f;
    if (p.reset && !second && p.reset[row] != 0.f) h = 0.f;                 // init_rnn_states_item (rnn.py:86-92)
    hs[(size_t)row * GH + j] = h;                                           // slot 0
    const float* gi = gi_base + (size_t)row * p.ld_gi;
    float g_r = gi[j], g_z = gi[GH + j], g_n = gi[2 * GH + j];
    for (int t = 0; t < p.T1; ++t) {
        // next step's input-side gates do not depend on h: issue their loads before the dot products
        float nx_r = 0.f, nx_z = 0.f, nx_n = 0.f;
        if (t + 1 < p.T1) {
            const float* g2 = gi + (size_t)(t + 1) * R * p.ld_gi;
            nx_r = g2[j]; nx_z = g2[GH + j]; nx_n = g2[2 * GH + j];
        }
        hl[j] = h;
        lds_barrier();                      // LDS only: __syncthreads() would also drain the global stores / prefetch loads
        float ar = br, az = bz, an = bn;
#pragma unroll
        for (int q = 0; q < GH / 4; ++q) {
            const float4 hv = reinterpret_cast<const float4*>(hl)[q];       // same address in every lane: broadcast
            ar = fmaf(wr[4 * q], hv.x, ar); ar = fmaf(wr[4 * q + 1], hv.y, ar);
            ar = fmaf(wr[4 * q + 2], hv.z, ar); ar = fmaf(wr[4 * q + 3], hv.w, ar);
            az = fmaf(wz[4 * q], hv.x, az); az = fmaf(wz[4 * q + 1], hv.y, az);
            az = fmaf(wz[4 * q + 2], hv.z, az); az = fmaf(wz[4 * q + 3], hv.w, az);
            an = fmaf(wn[4 * q], hv.x, an); an = fmaf(wn[4 * q + 1], hv.y, an);
            an = fmaf(wn[4 * q + 2], hv.z, an); an = fmaf(wn[4 * q + 3], hv.w, an);
        }
        lds_barrier();                                                      // hl is rewritten next step
        const float r = sigmoid_f(g_r + ar);
        const float z = sigmoid_f(g_z + az);
        const float n = tanhf(g_n + r * an);
        h = (h - n) * z + n;
        const size_t o = (size_t)t * R + row;
        hs[((size_t)(t + 1) * R + row) * GH + j] = h;
        if (gates) {
            float* g = gates + o * 4 * GH;
            g[j] = r; g[GH + j] = z; g[2 * GH + j] = n; g[3 * GH + j] = an;
        }
        g_r = nx_r; g_z = nx_z; g_n = nx_n;
    }
    if (p.h_last && !second) p.h_last[(size_t)row * GH + j] = h;
}

// BPTT.  Lane k owns hidden unit k: column k of W_hh (192 values) in registers.
__global__ void __launch_bounds__(64) gru_backward_kernel(xrl_gru_bwd_t p) {
    __shared__ __attribute__((aligned(16))) float gl[3 * GH];
    const int row = blockIdx.x, k = threadIdx.x;
    const int R = p.R;
    float wc[3 * GH];
#pragma unroll
    for (int jj = 0; jj < 3 * GH; ++jj) wc[jj] = p.w_hh[(size_t)jj * GH + k];
    float carry = 0.f;
    // operands of step t are loaded one step ahead (they do not depend on the carry)
    size_t o = (size_t)(p.T1 - 1) * R + row;
    const float* g = p.gates + o * 4 * GH;
    float r = g[k], z = g[GH + k], n = g[2 * GH + k], hn = g[3 * GH + k];
    float hp = p.hs[o * GH + k], dhs = p.d_hs[o * p.ld_dhs + k];            // hs slot t = h_{t-1}
    for (int t = p.T1 - 1; t >= 0; --t) {
        float r2 = 0.f, z2 = 0.f, n2 = 0.f, hn2 = 0.f, hp2 = 0.f, dhs2 = 0.f;
        if (t > 0) {
            const size_t o2 = o - R;
            const float* g2 = p.gates + o2 * 4 * GH;
            r2 = g2[k]; z2 = g2[GH + k]; n2 = g2[2 * GH + k]; hn2 = g2[3 * GH + k];
            hp2 = p.hs[o2 * GH + k]; dhs2 = p.d_hs[o2 * p.ld_dhs + k];
        }
        const float dh = dhs + carry;
        const float dn_pre = dh * (1.f - z) * (1.f - n * n);
        const float dz_pre = dh * (hp - n) * z * (1.f - z);
        const float dr_pre = dn_pre * hn * r * (1.f - r);
        const float dhn = dn_pre * r;
        float* dgi = p.d_gi + o * p.ld_dgi;
        dgi[k] = dr_pre; dgi[GH + k] = dz_pre; dgi[2 * GH + k] = dn_pre;
        float* dgh = p.d_gh + o * 3 * GH;
        dgh[k] = dr_pre; dgh[GH + k] = dz_pre; dgh[2 * GH + k] = dhn;
        gl[k] = dr_pre; gl[GH + k] = dz_pre; gl[2 * GH + k] = dhn;
        lds_barrier();
        float acc = dh * z;
#pragma unroll
        for (int q = 0; q < 3 * GH / 4; ++q) {
            const float4 gv = reinterpret_cast<const float4*>(gl)[q];
            acc = fmaf(gv.x, wc[4 * q], acc); acc = fmaf(gv.y, wc[4 * q + 1], acc);
            acc = fmaf(gv.z, wc[4 * q + 2], acc); acc = fmaf(gv.w, wc[4 * q + 3], acc);
        }
        lds_barrier();
        carry = acc;
        r = r2; z = z2; n = n2; hn = hn2; hp = hp2; dhs = dhs2;
        o -= R;
    }
    if (p.d_h0) p.d_h0[(size_t)row * GH + k] = carry;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_gru_forward(const xrl_gru_fwd_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->gi && p->w_hh && p->b_hh && p->hs);
    XRL_CHECK_ARG(p->H == GH);                       // one lane per hidden unit (3m.yaml: recurrent_hidden_size 64)
    XRL_CHECK_ARG(p->R > 0 && p->T1 > 0 && p->ld_gi >= 3 * GH);
    const bool dual = p->gi2 != nullptr;
    XRL_CHECK_ARG(!dual || (p->w_hh2 && p->b_hh2 && p->hs2));
    hipLaunchKernelGGL(gru_forward_kernel, dim3(dual ? 2 * p->R : p->R), dim3(64), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_gru_backward(const xrl_gru_bwd_t* p, xrl_stream_t stream) {
    XRL_CHECK_ARG(p && p->d_hs && p->hs && p->gates && p->w_hh && p->d_gi && p->d_gh);
    XRL_CHECK_ARG(p->H == GH);
    XRL_CHECK_ARG(p->R > 0 && p->T1 > 0 && p->ld_dhs >= GH && p->ld_dgi >= 3 * GH);
    hipLaunchKernelGGL(gru_backward_kernel, dim3(p->R), dim3(64), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
