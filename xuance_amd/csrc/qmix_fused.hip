// QMIX_Learner.update for feed-forward agents (multi_agent_rl/qmix_learner.py:24-112; heads/q_mix_head.py:28-95;
// value_factorization.py:66-150) as ONE launch: per-agent Q networks (eval on obs, eval + target on next_obs), masked
// double-Q target action, the eval / target hyper-networks, monotonic mixing, TD error, and the whole backward pass down
// to every weight gradient.
//
// Shape of the problem: 32 transitions x 3 agents, networks 30-64-64-9 and 48-{32,..}-{96,32,1}: ~9 MFLOP and 69 KB of
// parameters per update.  The layered path spends 68 us on it in 9 launches, every one of them latency-bound.  Nothing in
// the update couples two transitions except the constant 1/B of the mean, so the batch is cut into groups of
// `items_per_wg` transitions and a workgroup carries its group through everything with all activations in LDS ("per-agent
// Q in LDS"): no grid barrier, no atomics; workgroup g writes its weight-gradient partial to slab g and xrl_reduce_adam
// sums the slabs in fixed order.  At 12..24 rows per workgroup an MFMA tile would be mostly padding and the chain is
// latency-bound either way, so the products are plain FMA loops over LDS rows.  What decides the time is the number of
// DEPENDENT global round trips: a first version that read each layer's weights from L2 inside its product made ~30 of them
// (75 us, slower than the layered path).  So the weights come to LDS too, in two bursts with every load in flight at once:
// [eval agent | target agent | target mixer] at the start, and the eval mixer over the target mixer's space once the
// target hyper-networks have run (106 KB of weights + ~30 KB of activations at 4 transitions per workgroup).
#include "common.h"
#include "rng.h"

// (the library is built with -ffp-contract=off for the kernels that must round like NumPy; nothing here has to, and the
// products are VALU-bound: fused multiply-adds halve their instruction count)
#pragma clang fp contract(fast)

namespace xrl {

constexpr int QF_THREADS = 1024;                // 16 waves: independent products run side by side on their own thread ranges
constexpr int QF_PAD = 4;                     // LDS rows are allocated in multiples of 4 (the 4-row products read whole groups)

__device__ __forceinline__ float qf_elu(float x) { return x > 0.f ? x : expm1f(x); }

// The launch runs ~40 small matrix products ONCE each on a handful of workgroups, and the instruction cache is cold at every
// launch: with the product routines inlined at every call site the kernel was 70 KB of straight-line code and spent its
// time fetching instructions (190 k cycles, every phase 10-20x its arithmetic).  The routines are therefore real functions
// (one copy each, reused by every product); their LDS operands are offsets into the one dynamic LDS block so that the
// accesses stay ds_read / ds_write.
extern __shared__ __attribute__((aligned(16))) float qf_lds[];
typedef const __attribute__((address_space(1))) float* QfGlobalIn;      // (a generic pointer argument would mean flat accesses)
typedef __attribute__((address_space(1))) float* QfGlobalOut;
typedef float qf_f4 __attribute__((ext_vector_type(4)));

// n4 float4 from a weight image in global memory -> LDS offset dst: the images have the LDS layout already (rows padded to
// pad4(K) + 4 floats, see xrl_qmix_fused_layout), so a stage is a straight copy with eight loads in flight per thread and no
// index arithmetic (a first version re-laid the nn.Linear matrices on the way: one exposed memory latency per matrix and a
// division per element made the start-up burst 34 k cycles)
__device__ __noinline__ void qf_copy(QfGlobalIn src, int dst, int n4) {
    float* lds = qf_lds;
    const __attribute__((address_space(1))) qf_f4* g = reinterpret_cast<const __attribute__((address_space(1))) qf_f4*>(src);
    for (int i = threadIdx.x; i < n4; i += 8 * QF_THREADS) {
        qf_f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int q = i + j * QF_THREADS; v[j] = 0.f; if (q < n4) v[j] = g[q]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int q = i + j * QF_THREADS; if (q < n4) *reinterpret_cast<qf_f4*>(lds + dst + 4 * q) = v[j]; }
    }
}

// 16-byte load that is served by the L2, never by this CU's L1 (agent-scope `sc1`): what another workgroup of the SAME launch has
// written since this CU last read the line (xrl_qmix_fused_phase: weight images, gradient slabs)
__device__ __forceinline__ qf_f4 qf_ld4_dev(__amdgpu_buffer_rsrc_t rs, int q4) {
    typedef unsigned qf_u4 __attribute__((ext_vector_type(4)));
    const qf_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, q4 * 16, 0, 16);           // aux 16 = sc1
    qf_f4 r; r.x = __uint_as_float(v.x); r.y = __uint_as_float(v.y); r.z = __uint_as_float(v.z); r.w = __uint_as_float(v.w);
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t qf_rsrc(const float* p, long long floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}
__device__ __noinline__ void qf_copy_dev(const float* src, int src_floats, int dst, int n4) {     // qf_copy through qf_ld4_dev
    float* lds = qf_lds;
    const __amdgpu_buffer_rsrc_t rs = qf_rsrc(src, src_floats);
    for (int i = threadIdx.x; i < n4; i += 8 * QF_THREADS) {
        qf_f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int q = i + j * QF_THREADS; v[j] = 0.f; if (q < n4) v[j] = qf_ld4_dev(rs, q); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int q = i + j * QF_THREADS; if (q < n4) *reinterpret_cast<qf_f4*>(lds + dst + 4 * q) = v[j]; }
    }
}

// act_apply's switch is if-converted by hipcc (tanhf AND expf evaluated for every value, ~775 cycles each: common.h): the activations
// these networks actually use get a branch of their own (act is uniform), the rest goes through a real call
__device__ __noinline__ float qf_act_slow(float v, int act) { return act_apply(v, act); }
__device__ __forceinline__ float qf_act(float v, int act) {
    if (act == XRL_ACT_NONE) return v;
    if (act == XRL_ACT_RELU) return v > 0.f ? v : 0.f;
    return qf_act_slow(v, act);
}

// FAST: the activation is none / relu (what these networks use), applied inline -- the product routine is then a leaf function.  With
// the call to qf_act_slow inside, qf_lin_fwd_fn kept a value in a callee-saved VGPR across it and so saved / restored that register
// through SCRATCH memory in its own prologue / epilogue: a scratch reload in front of every return, ~10 times per launch per wave.
template <int QF_RB, bool FAST>
__device__ __forceinline__ void qf_lin_fwd_t(int W, int ldw, int b, int K, int Nout, int in, int ldi, int rows, int out, int ldo, int act,
                                             int tid0) {
    float* lds = qf_lds;
    const int n_rg = (rows + QF_RB - 1) / QF_RB, K4 = (K + 3) & ~3;
    for (int item = (threadIdx.x - tid0) & (QF_THREADS - 1); item < Nout * n_rg; item += QF_THREADS) {
        const int n = item % Nout, r0 = (item / Nout) * QF_RB;
        const float* w = lds + W + n * ldw;
        const float* x0 = lds + in + r0 * ldi;
        float acc[QF_RB];
#pragma unroll
        for (int j = 0; j < QF_RB; ++j) acc[j] = 0.f;
#pragma unroll(QF_RB == 1 ? 4 : 1)
        for (int k = 0; k < K4; k += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
            for (int j = 0; j < QF_RB; ++j) {
                const float4 x = *reinterpret_cast<const float4*>(x0 + j * ldi + k);   // rows padded: readable
                acc[j] = __builtin_fmaf(x.x, wv.x, acc[j]); acc[j] = __builtin_fmaf(x.y, wv.y, acc[j]);
                acc[j] = __builtin_fmaf(x.z, wv.z, acc[j]); acc[j] = __builtin_fmaf(x.w, wv.w, acc[j]);
            }
        }
        const float bias = lds[b + n];
#pragma unroll
        for (int j = 0; j < QF_RB; ++j)
            if (r0 + j < rows) {
                const float v = acc[j] + bias;
                lds[out + (r0 + j) * ldo + n] = FAST ? ((act == XRL_ACT_RELU && !(v > 0.f)) ? 0.f : v) : qf_act(v, act);
            }
    }
}

// (few rows: one row per work item spreads the product over more threads; many rows: four rows share every weight read.
//  tid0, a multiple of 64: the thread that takes work item 0 -- products between two barriers get disjoint thread ranges)
__device__ __noinline__ void qf_lin_fwd_fn(int W, int ldw, int b, int K, int Nout, int in, int ldi, int rows, int out, int ldo, int act,
                                           int tid0) {
    if (Nout * rows > 512) qf_lin_fwd_t<4, true>(W, ldw, b, K, Nout, in, ldi, rows, out, ldo, act, tid0);
    else qf_lin_fwd_t<1, true>(W, ldw, b, K, Nout, in, ldi, rows, out, ldo, act, tid0);
}
__device__ __noinline__ void qf_lin_fwd_any_fn(int W, int ldw, int b, int K, int Nout, int in, int ldi, int rows, int out, int ldo, int act,
                                               int tid0) {      // (any other activation: through qf_act_slow)
    if (Nout * rows > 512) qf_lin_fwd_t<4, false>(W, ldw, b, K, Nout, in, ldi, rows, out, ldo, act, tid0);
    else qf_lin_fwd_t<1, false>(W, ldw, b, K, Nout, in, ldi, rows, out, ldo, act, tid0);
}

// does any lane of this wave get one of n_items work items when item 0 goes to thread tid0?  (a call costs every wave that
// makes it a few hundred cycles even when it finds nothing to do -- with ~45 products per launch that was most of the time)
__device__ __forceinline__ bool qf_wave_in(int n_items, int tid0) {
    return (int)(((threadIdx.x & ~63u) - (unsigned)tid0) & (QF_THREADS - 1)) < n_items;
}
// ANYACT: the kernel instance may meet an activation other than none / relu.  A template argument of the kernels (round 6): an instance
// that never does (the networks of configs/qmix/sc2/3m.yaml, every fixture) then has no call site of the twin whose callee-saved register
// goes through scratch -- and with it no scratch allocation at all (16 bytes per lane were reserved for a routine the launch never entered)
template <bool ANYACT>
__device__ __forceinline__ void qf_lin_fwd(int W, int ldw, int b, int K, int Nout, int in, int ldi, int rows, int out, int ldo, int act,
                                           int tid0) {
    if (qf_wave_in(Nout * rows, tid0)) {
        if (!ANYACT || act == XRL_ACT_NONE || act == XRL_ACT_RELU) qf_lin_fwd_fn(W, ldw, b, K, Nout, in, ldi, rows, out, ldo, act, tid0);
        else qf_lin_fwd_any_fn(W, ldw, b, K, Nout, in, ldi, rows, out, ldo, act, tid0);
    }
}

// dx[r][k] = (sum_n dz[r][n] W[n][k]) * act'(y[r][k])      (y = the layer input's own activation output, < 0: none)
template <int QF_RB>
__device__ __forceinline__ void qf_lin_bwd_data_t(int W, int ldw, int K, int Nout, int dz, int ldz, int rows, int dx, int ldx, int y,
                                                  int ldy, int act, int tid0) {
    float* lds = qf_lds;
    const int n_rg = (rows + QF_RB - 1) / QF_RB;
    for (int item = (threadIdx.x - tid0) & (QF_THREADS - 1); item < K * n_rg; item += QF_THREADS) {
        const int k = item % K, r0 = (item / K) * QF_RB;
        float acc[QF_RB];
#pragma unroll
        for (int j = 0; j < QF_RB; ++j) acc[j] = 0.f;
#pragma unroll(QF_RB == 1 ? 8 : 2)
        for (int n = 0; n < Nout; ++n) {
            const float wv = lds[W + n * ldw + k];
#pragma unroll
            for (int j = 0; j < QF_RB; ++j) acc[j] += lds[dz + (r0 + j) * ldz + n] * wv;
        }
#pragma unroll
        for (int j = 0; j < QF_RB; ++j)
            if (r0 + j < rows) lds[dx + (r0 + j) * ldx + k] = acc[j] * (y >= 0 ? act_grad_from_out(lds[y + (r0 + j) * ldy + k], act) : 1.f);
    }
}

__device__ void qf_lin_bwd_data_fn(int W, int ldw, int K, int Nout, int dz, int ldz, int rows, int dx, int ldx, int y, int ldy, int act,
                                   int tid0);
__device__ __forceinline__ void qf_lin_bwd_data(int W, int ldw, int K, int Nout, int dz, int ldz, int rows, int dx, int ldx, int y,
                                                int ldy, int act, int tid0) {
    if (qf_wave_in(K * rows, tid0)) qf_lin_bwd_data_fn(W, ldw, K, Nout, dz, ldz, rows, dx, ldx, y, ldy, act, tid0);
}
__device__ __noinline__ void qf_lin_bwd_data_fn(int W, int ldw, int K, int Nout, int dz, int ldz, int rows, int dx, int ldx, int y,
                                                int ldy, int act, int tid0) {
    if (K * rows > 512) qf_lin_bwd_data_t<4>(W, ldw, K, Nout, dz, ldz, rows, dx, ldx, y, ldy, act, tid0);
    else qf_lin_bwd_data_t<1>(W, ldw, K, Nout, dz, ldz, rows, dx, ldx, y, ldy, act, tid0);
}

// dW[n][k] = sum_r dz[r][n] in[r][k],  db[n] = sum_r dz[r][n]   -> this workgroup's slab (every element written once)
__device__ void qf_lin_bwd_weight_fn(QfGlobalOut dW, QfGlobalOut db, int K, int Nout, int dz, int ldz, int in, int ldi, int rows, int tid0);
__device__ __forceinline__ void qf_lin_bwd_weight(QfGlobalOut dW, QfGlobalOut db, int K, int Nout, int dz, int ldz, int in, int ldi,
                                                  int rows, int tid0) {
    if (qf_wave_in(Nout * ((K + 3) / 4), tid0) || qf_wave_in(Nout, tid0 + 512))
        qf_lin_bwd_weight_fn(dW, db, K, Nout, dz, ldz, in, ldi, rows, tid0);
}
__device__ __noinline__ void qf_lin_bwd_weight_fn(QfGlobalOut dW, QfGlobalOut db, int K, int Nout, int dz, int ldz, int in, int ldi,
                                                  int rows, int tid0) {
    float* lds = qf_lds;
    const int kq = (K + 3) / 4;
    const bool vec = (K & 3) == 0 && (((uintptr_t)dW & 15) == 0);
    for (int item = (threadIdx.x - tid0) & (QF_THREADS - 1); item < Nout * kq; item += QF_THREADS) {
        const int q = item % kq, n = item / kq;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int r = 0; r < rows; ++r) {
            const float g = lds[dz + r * ldz + n];
            const float4 x = *reinterpret_cast<const float4*>(lds + in + r * ldi + 4 * q);
            acc.x += g * x.x; acc.y += g * x.y; acc.z += g * x.z; acc.w += g * x.w;
        }
        QfGlobalOut o = dW + (size_t)n * K + 4 * q;
        if (vec) { qf_f4 t; t.x = acc.x; t.y = acc.y; t.z = acc.z; t.w = acc.w; *reinterpret_cast<__attribute__((address_space(1))) qf_f4*>(o) = t; }
        else {
            if (4 * q + 0 < K) o[0] = acc.x;
            if (4 * q + 1 < K) o[1] = acc.y;
            if (4 * q + 2 < K) o[2] = acc.z;
            if (4 * q + 3 < K) o[3] = acc.w;
        }
    }
    for (int n = (threadIdx.x - tid0 - 512) & (QF_THREADS - 1); n < Nout; n += QF_THREADS) {     // (other threads than item 0's)
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += lds[dz + r * ldz + n];
        db[n] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same three products on the matrix cores (v_mfma_f32_16x16x4_f32), for workgroups that carry enough transitions to fill a
// 16-row tile (items_per_wg * N >= 8: `products`).  Everything is computed TRANSPOSED -- D[feature][row] = sum A[feature][.] B[.][row]
// -- because then both operands and the result are 16-byte LDS accesses in the row-major [row][feature] layout the VALU products
// use: lane (g = lane / 16, cl = lane % 16) supplies A[m = cl][k = 4 g + s] and B[k = 4 g + s][n = cl] to the MFMA with component s
// of a float4, and holds D[m = 4 g + i][n = cl] in element i of the result.  A work item is one (16-row, 16-feature) tile; item j of
// a product goes to wave (tid0 / 64 + j) mod 16 -- the call sites hand consecutive products consecutive tile ranges, so a phase's
// tiles are dealt round-robin to the 16 waves.  Out-of-range rows / features: the WEIGHT-side operand is zeroed (its image holds
// other matrices there), the activation-side operand is read at a clamped address (finite values times zero).
// A VALU product phase at 3 rows per workgroup and an MFMA phase at 15 rows cost about the same (~1 k cycles: one LDS latency +
// K / 4 dependent MFMAs); what the larger groups buy is 5x fewer workgroups, slabs and weight bursts per update.
typedef float qf_acc4 __attribute__((ext_vector_type(4)));
#define QF_MFMA(a, b, acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), acc, 0, 0, 0)

__device__ __forceinline__ bool qf_mm_wave_in(int n_tiles, int tid0) {
    return (int)((((threadIdx.x & ~63u) - (unsigned)tid0) & (QF_THREADS - 1)) >> 6) < n_tiles;
}
__host__ __device__ inline int qf_tiles(int a, int b) { return ((a + 15) >> 4) * ((b + 15) >> 4); }

// (Arguments of a real function arrive in vector registers and the compiler must assume they differ per lane: every loop and
//  branch on them becomes exec-mask juggling, every division a 40-instruction expansion -- measured 3 k cycles for a tile of 8
//  MFMAs.  They ARE uniform: readfirstlane hands them to the scalar unit.)
#define QF_UNI(x) x = __builtin_amdgcn_readfirstlane(x)

// element-wise activation of a tile's four values; act is uniform, so this is one scalar branch
__device__ __forceinline__ void qf_act4(float (&v)[4], int act) {
    if (act == XRL_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    } else if (act != XRL_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = qf_act_slow(v[i], act);
    }
}

// out[r][n] = act(sum_k in[r][k] W[n][k] + b[n])
__device__ __noinline__ void qf_mm_fwd_fn(int W, int ldw, int b, int K, int Nout, int in, int ldi, int rows, int out, int ldo, int act,
                                          int tid0) {
    QF_UNI(W); QF_UNI(ldw); QF_UNI(b); QF_UNI(K); QF_UNI(Nout); QF_UNI(in); QF_UNI(ldi); QF_UNI(rows); QF_UNI(out); QF_UNI(ldo);
    QF_UNI(act); QF_UNI(tid0);
    float* lds = qf_lds;
    const int lane = threadIdx.x & 63, cl = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(((threadIdx.x - (unsigned)tid0) & (QF_THREADS - 1)) >> 6));
    const int n_nt = (Nout + 15) >> 4, n_items = ((rows + 15) >> 4) * n_nt, K4 = (K + 3) & ~3, kc = (K + 15) >> 4;
    for (int item = wv; item < n_items; item += QF_THREADS / 64) {
        const int nt = item % n_nt, rt = item / n_nt;
        const int n = nt * 16 + cl, r = min(rt * 16 + cl, rows - 1);
        const bool n_ok = n < Nout;
        const float* wrow = lds + W + min(n, Nout - 1) * ldw;
        const float* xrow = lds + in + r * ldi;
        qf_acc4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < kc; c0 += 4) {               // up to four k-chunks: all their LDS reads in flight, then 16 MFMAs on four chains
            float4 a[4], x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = 16 * (c0 + j) + 4 * g, kr = min(kk, K4 - 4);
                a[j] = *reinterpret_cast<const float4*>(wrow + kr);
                x[j] = *reinterpret_cast<const float4*>(xrow + kr);
                if (!n_ok || kk >= K4) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c0 + j < kc) { QF_MFMA(a[j].x, x[j].x, acc0); QF_MFMA(a[j].y, x[j].y, acc1); QF_MFMA(a[j].z, x[j].z, acc2); QF_MFMA(a[j].w, x[j].w, acc3); }
            }
        }
        const int n0 = nt * 16 + 4 * g, row = rt * 16 + cl;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ((acc0[i] + acc1[i]) + (acc2[i] + acc3[i])) + lds[b + min(n0 + i, Nout - 1)];
        qf_act4(v, act);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (n0 + i >= Nout) v[i] = 0.f;
        if (row < rows && n0 < ((Nout + 3) & ~3)) *reinterpret_cast<float4*>(lds + out + row * ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// dx[r][k] = (sum_n dz[r][n] W[n][k]) * act'(y[r][k])
__device__ __noinline__ void qf_mm_bwd_data_fn(int W, int ldw, int K, int Nout, int dz, int ldz, int rows, int dx, int ldx, int y, int ldy,
                                               int act, int tid0) {
    QF_UNI(W); QF_UNI(ldw); QF_UNI(K); QF_UNI(Nout); QF_UNI(dz); QF_UNI(ldz); QF_UNI(rows); QF_UNI(dx); QF_UNI(ldx); QF_UNI(y); QF_UNI(ldy);
    QF_UNI(act); QF_UNI(tid0);
    float* lds = qf_lds;
    const int lane = threadIdx.x & 63, cl = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(((threadIdx.x - (unsigned)tid0) & (QF_THREADS - 1)) >> 6));
    const int n_kt = (K + 15) >> 4, n_items = ((rows + 15) >> 4) * n_kt, N4 = (Nout + 3) & ~3, nc = (Nout + 15) >> 4;
    for (int item = wv; item < n_items; item += QF_THREADS / 64) {
        const int kt = item % n_kt, rt = item / n_kt;
        const int k = kt * 16 + cl, r = min(rt * 16 + cl, rows - 1);
        const bool k_ok = k < K;
        const float* wcol = lds + W + min(k, K - 1);                      // A[m = k][contraction n] = W[n][k]
        const float* zrow = lds + dz + r * ldz;
        qf_acc4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < nc; c0 += 2) {
            float4 z[2];
            float a[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nn = 16 * (c0 + j) + 4 * g;
                z[j] = *reinterpret_cast<const float4*>(zrow + min(nn, N4 - 4));
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const float w = wcol[min(nn + s_, Nout - 1) * ldw];
                    a[j][s_] = (k_ok && nn + s_ < Nout) ? w : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (c0 + j < nc) { QF_MFMA(a[j][0], z[j].x, acc0); QF_MFMA(a[j][1], z[j].y, acc1); QF_MFMA(a[j][2], z[j].z, acc2); QF_MFMA(a[j][3], z[j].w, acc3); }
            }
        }
        const int k0 = kt * 16 + 4 * g, row = rt * 16 + cl;
        if (row < rows && k0 < ((K + 3) & ~3)) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = y >= 0 ? act_grad_from_out(lds[y + row * ldy + min(k0 + i, K - 1)], act) : 1.f;
                v[i] = k0 + i < K ? ((acc0[i] + acc1[i]) + (acc2[i] + acc3[i])) * d : 0.f;
            }
            *reinterpret_cast<float4*>(lds + dx + row * ldx + k0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// dW[n][k] = sum_r dz[r][n] in[r][k],  db[n] = sum_r dz[r][n]   -> this workgroup's slab
__device__ __noinline__ void qf_mm_bwd_weight_fn(QfGlobalOut dW, QfGlobalOut db, int K, int Nout, int dz, int ldz, int in, int ldi, int rows,
                                                 int tid0) {
    QF_UNI(K); QF_UNI(Nout); QF_UNI(dz); QF_UNI(ldz); QF_UNI(in); QF_UNI(ldi); QF_UNI(rows); QF_UNI(tid0);
    float* lds = qf_lds;
    const int lane = threadIdx.x & 63, cl = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(((threadIdx.x - (unsigned)tid0) & (QF_THREADS - 1)) >> 6));
    const int n_kt = (K + 15) >> 4, n_items = ((Nout + 15) >> 4) * n_kt, rc = (rows + 15) >> 4;
    for (int item = wv; item < n_items; item += QF_THREADS / 64) {
        const int kt = item % n_kt, nt = item / n_kt;
        const int n = nt * 16 + cl, k = kt * 16 + cl;
        const float* zcol = lds + dz + min(n, Nout - 1);                  // A[m = n][contraction r] = dz[r][n]
        const float* xcol = lds + in + min(k, K - 1);                     // B[contraction r][n' = k] = in[r][k]
        qf_acc4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < rc; ++c) {
            float a[4], x[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int r = 16 * c + 4 * g + s_, rr = min(r, rows - 1);
                const float zv = zcol[rr * ldz];
                a[s_] = (n < Nout && r < rows) ? zv : 0.f;
                x[s_] = xcol[rr * ldi];
            }
            QF_MFMA(a[0], x[0], acc0); QF_MFMA(a[1], x[1], acc1); QF_MFMA(a[2], x[2], acc0); QF_MFMA(a[3], x[3], acc1);
            if (kt == 0) { QF_MFMA(a[0], 1.f, accb); QF_MFMA(a[1], 1.f, accb); QF_MFMA(a[2], 1.f, accb); QF_MFMA(a[3], 1.f, accb); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nn = nt * 16 + 4 * g + i;
            if (nn < Nout && k < K) dW[(size_t)nn * K + k] = acc0[i] + acc1[i];
            if (kt == 0 && cl == 0 && nn < Nout) db[nn] = accb[i];
        }
    }
}

struct QfLds {                      // offsets (floats) into the dynamic LDS block; host and device compute them alike
    int x0, x1, h[XRL_QF_MAX_LAYERS], q, qne, qnt, t0, t1, u0, u1, s0, s1, hid_e, raw_e, hid_t, raw_t, d_raw, d_hid, total;
    int ld[XRL_QF_MAX_LAYERS + 1], ldmax, lds, ldh, ldr, rows_pad;
    int we[XRL_QF_MAX_LAYERS], be[XRL_QF_MAX_LAYERS], wt[XRL_QF_MAX_LAYERS], bt[XRL_QF_MAX_LAYERS], ldw[XRL_QF_MAX_LAYERS];
    int mw[5], mb[5], mldw[5];      // the mixer staged at the moment (target first, then eval): FIRST, B1, W1, W2, B2
    int act_i, rew, term, amask, avail, clear_end;
    int ag_e, ag_t, mix;            // LDS offsets of the three weight blocks (each the copy of one image block)
};

__host__ __device__ inline int qf_pad4(int w) { return (w + 3) / 4 * 4; }
__host__ __device__ inline bool qf_mfma_products(const xrl_qmix_fused_t& p) {
    return p.products == 1 || (p.products == 0 && p.items_per_wg * p.N >= 8);
}

// the weight images: agent block [W_0 .. W_{L-1} | b_0 .. b_{L-1}], mixer block [FIRST B1 W1 W2 B2 | their biases]; matrix
// rows padded to pad4(K) + 4 floats (conflict-free 16-byte LDS reads with one row per lane), bias vectors to quads
__host__ __device__ inline void qf_image_layout(const xrl_qmix_fused_t& p, xrl_qf_image_t& im) {
    int off = 0;
    for (int l = 0; l < XRL_QF_MAX_LAYERS; ++l) { im.w[l] = im.b[l] = im.ldw[l] = 0; }
    for (int l = 0; l < p.n_layers; ++l) { im.ldw[l] = qf_pad4(p.dims[l]) + 4; im.w[l] = off; off += p.dims[l + 1] * im.ldw[l]; }
    for (int l = 0; l < p.n_layers; ++l) { im.b[l] = off; off += qf_pad4(p.dims[l + 1]); }
    im.agent_floats = off;
    const int mK[5] = {p.S, p.S, p.HH, p.HH, p.HH}, mN[5] = {3 * p.HH, p.H, p.N * p.H, p.H, 1};
    off = 0;
    for (int i = 0; i < 5; ++i) { im.mldw[i] = qf_pad4(mK[i]) + 4; im.mw[i] = off; off += mN[i] * im.mldw[i]; }
    for (int i = 0; i < 5; ++i) { im.mb[i] = off; off += qf_pad4(mN[i]); }
    im.mixer_floats = off;
}

__host__ __device__ inline QfLds qf_layout(const xrl_qmix_fused_t& p) {
    QfLds L;
    const int rows = p.items_per_wg * p.N;
    // the 4-row VALU products read whole row groups; the matrix-core tiles clamp their row addresses instead (no padding rows:
    // 5 transitions x 3 agents = 15 rows per workgroup fit beside the weights, 16 padded ones would not)
    const int pad = qf_mfma_products(p) ? 1 : QF_PAD;
    L.rows_pad = (rows + pad - 1) / pad * pad;
    const int bw_pad = (p.items_per_wg + pad - 1) / pad * pad;
    int off = 0, ldmax = 0;
    // matrix-core tiles read one activation ROW per lane (16 rows at once): a row stride that is a multiple of 32 floats puts all
    // 16 on the same four LDS banks (measured: product phases of 6.4 k cycles instead of ~2 k) -> four floats of padding per row,
    // as the weight images have
    const int rpad = qf_mfma_products(p) ? 4 : 0;
    for (int l = 0; l <= p.n_layers; ++l) { L.ld[l] = qf_pad4(p.dims[l]) + rpad; if (l > 0 && L.ld[l] > ldmax) ldmax = L.ld[l]; }
    L.ldmax = ldmax;
    L.x0 = off; off += L.rows_pad * L.ld[0];
    L.x1 = off; off += L.rows_pad * L.ld[0];
    for (int l = 1; l < p.n_layers; ++l) { L.h[l] = off; off += L.rows_pad * L.ld[l]; }
    const int ldq = L.ld[p.n_layers];
    L.q = off; off += L.rows_pad * ldq;
    L.qne = off; off += L.rows_pad * ldq;
    L.qnt = off; off += L.rows_pad * ldq;
    L.t0 = off; off += L.rows_pad * ldmax;
    L.t1 = off; off += L.rows_pad * ldmax;
    L.u0 = off; off += L.rows_pad * ldmax;
    L.u1 = off; off += L.rows_pad * ldmax;
    L.lds = qf_pad4(p.S) + rpad; L.ldh = qf_pad4(3 * p.HH + p.H) + rpad; L.ldr = qf_pad4(p.N * p.H + p.H + 1) + rpad;
    L.s0 = off; off += bw_pad * L.lds;
    L.s1 = off; off += bw_pad * L.lds;
    L.hid_e = off; off += bw_pad * L.ldh;
    L.raw_e = off; off += bw_pad * L.ldr;
    L.hid_t = off; off += bw_pad * L.ldh;
    L.raw_t = off; off += bw_pad * L.ldr;
    L.d_raw = off; off += bw_pad * L.ldr;
    L.d_hid = off; off += bw_pad * L.ldh;
    L.act_i = off; off += L.rows_pad;
    L.rew = off; off += L.rows_pad;
    L.term = off; off += L.rows_pad;
    L.amask = off; off += L.rows_pad;
    L.avail = off; off += L.rows_pad * qf_pad4(p.A);
    L.clear_end = off;                                          // below: whole copies of image blocks (their padding is zero)
    xrl_qf_image_t im;
    qf_image_layout(p, im);
    L.ag_e = off; off += im.agent_floats;
    L.ag_t = off; off += im.agent_floats;
    L.mix = off; off += im.mixer_floats;
    for (int l = 0; l < p.n_layers; ++l) {
        L.ldw[l] = im.ldw[l];
        L.we[l] = L.ag_e + im.w[l]; L.be[l] = L.ag_e + im.b[l];
        L.wt[l] = L.ag_t + im.w[l]; L.bt[l] = L.ag_t + im.b[l];
    }
    for (int i = 0; i < 5; ++i) { L.mw[i] = L.mix + im.mw[i]; L.mb[i] = L.mix + im.mb[i]; L.mldw[i] = im.mldw[i]; }
    L.total = off;
    return L;
}

struct QfArgs { xrl_qmix_fused_t p; QfLds L; int agent4, mixer4, pad[2]; };   // L = qf_layout(p), block sizes in float4: from the host
struct QfArgsPh { QfArgs a; xrl_qmix_phase_t ph; };                            // the phase launch (xrl_qmix_fused_phase): + the optimiser's side
template <bool PHASE> struct qf_kernel_arg { typedef QfArgs type; };
template <> struct qf_kernel_arg<true> { typedef QfArgsPh type; };
typedef const __attribute__((address_space(4))) xrl_qmix_phase_t* QfPh;
// words of xrl_qmix_phase_t.sync: [2] time-out, [3] XCC mask of the launch, [8 + g] / [72 + g] the two meeting flags of workgroup g
constexpr int QFS_FAIL = 2, QFS_MASK = 3, QFS_A = 8, QFS_B = 72, QF_PHASE_MAX_WG = 64;
typedef const __attribute__((address_space(4))) QfArgs QfArgsK;
typedef const __attribute__((address_space(4))) xrl_qmix_fused_t* QfP;
typedef const __attribute__((address_space(4))) QfLds* QfL;

// MM: the products on the matrix cores (qf_mm_*) instead of the VALU loops.  A template argument, not a run-time branch: with both forms
// behind `if (mm)` at every call site the default (VALU) launch carried the other form's call sites through its cold instruction cache
// and ran 2.8 us slower (round 4 -> 5: 34.8 -> 37.6 us per update on every box, profiles/r0[2-5]_*_bench.json)
// ------------------------------------------------------------------------------------------------------------------
// xrl_qmix_fused_phase: the optimiser step of update u, done between two updates by the resident workgroups of the launch -- what
// xrl_reduce_adam does in a launch of its own: the same statements on the same elements in the same order (no clipping: nothing
// here needs the global norm, so the workgroups meet only to hand over slabs and parameters).  Returns whether the launch's
// workgroups sit on more than one XCD (then the meetings use agent-scope fences; inside one L2 drained plain stores + L1-bypassing
// loads are coherent).
typedef const __attribute__((address_space(4))) xrl_qmix_fused_t* QfPq;
typedef const __attribute__((address_space(4))) QfLds* QfLq;
__device__ __noinline__ void qf_phase_meet(unsigned* sync, int base, int wg, int n_wg, unsigned epoch, bool multi) {
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 64) {
        if (tid == 0) {
            if (multi) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_store(sync + base + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        for (;;) {
            const unsigned f = tid < n_wg ? __hip_atomic_load(sync + base + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all(f == epoch)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255) == 0 && (spins > 4000000 || __hip_atomic_load(sync + QFS_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (tid == 0) __hip_atomic_store(sync + QFS_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (multi) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __noinline__ bool qf_phase_step(QfPh ph, QfPq p, QfLq L, int wg, int n_wg, int u, int n_upd, unsigned step0, unsigned sched0, bool multi) {
#pragma clang fp contract(off)      // (this file contracts its products; the optimiser's statements must round like csrc/optim.hip's: every multiply and add on its own)
    float* lds = qf_lds;
    const int tid = threadIdx.x;
    unsigned* sync = ph->sync;
    const unsigned step = step0 + (unsigned)u + 1u;                       // the optimiser step this update performs
    long long* dbg = (wg == 0 && tid == 0 && u == n_upd / 2) ? p->dbg : nullptr;     // diagnostics (tools/probe_qmix_phase.py): [64 + k]
#define QPS(k) do { if (dbg) dbg[64 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
    QPS(0);
    // ---- meeting A: every slab of update u is in the L2
    qf_phase_meet(sync, QFS_A, wg, n_wg, 2u * step, multi);
    QPS(1);
    if (u == 0) {                                                         // every workgroup's XCC bit is in (set before its first arrival)
        const unsigned mask = __hip_atomic_load(sync + QFS_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        multi = __popc(mask) != 1;
    }
    const bool dead = __hip_atomic_load(sync + QFS_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    // ---- this workgroup's quads: [q_lo, q_lo + n_q) of the P / 4, an even split; thread (pq, sg) sums slabs sg, sg + 4, ... of quad q_lo + pq
    const long long P4 = ph->P / 4;
    const int per = (int)((P4 + n_wg - 1) / n_wg), q_lo = wg * per, n_q = max(0, min(per, (int)P4 - q_lo));
    double (*gsum)[4] = reinterpret_cast<double (*)[4]>(lds + L->ag_e);   // [4][n_q][4] where the eval agent's weights were
    double* red = reinterpret_cast<double*>(lds + L->ag_e) + 16 * per;    // [16] partial squared norms
    const int n_split = n_wg;
    const __amdgpu_buffer_rsrc_t rs_s = qf_rsrc((const float*)p->slabs, (long long)n_split * p->slab_stride);
    const int st4 = (int)(p->slab_stride / 4);
    for (int it = tid; it < 4 * n_q; it += QF_THREADS) {
        const int sg = it / n_q, pq = it - sg * n_q, qi = q_lo + pq;
        double gx = 0.0, gy = 0.0, gz = 0.0, gw = 0.0;
        int sl = sg;
        for (; sl + 28 < n_split; sl += 32) {
            qf_f4 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = qf_ld4_dev(rs_s, (sl + 4 * j) * st4 + qi);
#pragma unroll
            for (int j = 0; j < 8; ++j) { gx += w[j].x; gy += w[j].y; gz += w[j].z; gw += w[j].w; }
        }
        for (; sl < n_split; sl += 4) { const qf_f4 w = qf_ld4_dev(rs_s, sl * st4 + qi); gx += w.x; gy += w.y; gz += w.z; gw += w.w; }
        double* o = gsum[sg * n_q + pq];
        o[0] = gx; o[1] = gy; o[2] = gz; o[3] = gw;
    }
    QPS(2);
    // the loss sums of update u (xrl_sum_partials' order, row by row) ride with the last workgroup: its rows are all in the L2 now
    // (eight L1-bypassing loads in flight per thread: as 32 dependent atomic loads this made the last workgroup the launch's straggler --
    //  every other workgroup stood 14 k cycles in meeting B)
    if (wg == n_wg - 1 && tid >= QF_THREADS - 8) {
        typedef unsigned qf_u2 __attribute__((ext_vector_type(2)));
        const int c = tid - (QF_THREADS - 8), B = p->B;
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ph->phase_partials + (size_t)u * B * 8), 0, B * 64, 0x00020000);
        double sacc = 0.0;
        int r = 0;
        for (; r + 8 <= B; r += 8) {
            qf_u2 w8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w8[q] = __builtin_amdgcn_raw_buffer_load_b64(rs_p, ((r + q) * 8 + c) * 8, 0, 16);
#pragma unroll
            for (int q = 0; q < 8; ++q) sacc += __hiloint2double((int)w8[q].y, (int)w8[q].x);
        }
        for (; r < B; ++r) { const qf_u2 w = __builtin_amdgcn_raw_buffer_load_b64(rs_p, (r * 8 + c) * 8, 0, 16); sacc += __hiloint2double((int)w.y, (int)w.x); }
        ph->epoch_sums[(size_t)u * 8 + c] = sacc;
    }
    const xrl_adam_state_t* st = ph->state;
    __syncthreads();
    QPS(3);
    // ---- Adam for the 4 n_q parameters (adam_step_kernel's statements; the step's float64 scalars -- lr / (1 - beta1^t), sqrt(1 - beta2^t):
    //      torch._single_tensor_adam -- come from qf_adam_scalars_kernel, launched in front of the phase: float64 pow() does not fit the
    //      128 registers a 1 024-thread workgroup leaves a thread)
    {
        const float step_size = ph->scalars[2 * u], bc2_sqrt = ph->scalars[2 * u + 1];
        const float eps = (float)st->eps, w1 = (float)(1.0 - st->beta1), fb2 = (float)st->beta2, w2 = (float)(1.0 - st->beta2), wd = (float)st->weight_decay;
        const bool hard = ph->sync_every > 0 && (int)step % ph->sync_every == 0;
        float* img_e = const_cast<float*>((const float*)p->img_eval);
        float* img_t = const_cast<float*>((const float*)p->img_target);
        double sq = 0.0;
        for (int e = tid; e < 4 * n_q && !dead; e += QF_THREADS) {
            const int pq = e >> 2, c = e & 3;
            const long long i = 4ll * (q_lo + pq) + c;
            const double t0 = ((gsum[0 * n_q + pq][c] + gsum[1 * n_q + pq][c]) + gsum[2 * n_q + pq][c]) + gsum[3 * n_q + pq][c];
            float g = (float)t0;                                          // rounded once (grad_reduce_kernel)
            sq += (double)g * (double)g;
            ph->grad[i] = g;                                              // (coef = 1: no clipping)
            const float p0 = ph->params[i], m0 = ph->m[i], v0 = ph->v[i];
            if (wd != 0.f) g += wd * p0;
            const float mi = m0 + (g - m0) * w1;
            const float vi = v0 * fb2 + w2 * g * g;
            ph->m[i] = mi; ph->v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            const float pn = p0 - step_size * (mi / denom);
            ph->params[i] = pn;
            const int j = ph->map[i];
            if (j >= 0) img_e[j] = pn;
            if (ph->act_image) { const int ja = ph->act_map[i]; if (ja >= 0) ph->act_image[ja] = pn; }
            if (hard) { ph->target[i] = pn; if (j >= 0) img_t[j] = pn; }
        }
        // this workgroup's share of the squared norm (reported, not used: no clipping)
        sq = wave_sum(sq);
        if ((tid & 63) == 0) red[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < QF_THREADS / 64; ++w) t += red[w];
            __hip_atomic_store(ph->sumsq_part + wg, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    QPS(4);
    // ---- meeting B: the new parameters / images are in the L2
    qf_phase_meet(sync, QFS_B, wg, n_wg, 2u * step + 1u, multi);
    QPS(5);
    if (wg == 0 && tid == 0) {                                            // (every workgroup took the state's counters before its first arrival)
        xrl_adam_state_t* sw = ph->state;
        double t = 0.0;
        for (int w = 0; w < n_wg; ++w) t += __hip_atomic_load(ph->sumsq_part + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool bad = __hip_atomic_load(sync + QFS_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        sw->last_grad_norm = bad ? __builtin_nan("") : sqrt(t);
        sw->step = (int)step;
        const int ns = (int)sched0 + u + 1;
        sw->sched_steps = ns;
        const int k2 = ns < sw->total_iters ? ns : sw->total_iters;
        sw->last_lr = sw->base_lr * (1.0 + (sw->end_factor - 1.0) * (double)k2 / (double)sw->total_iters);
        if (u == n_upd - 1 && ph->tick) *ph->tick += (unsigned)ph->tick_inc;
    }
#undef QPS
    return multi;
}

// step_size and sqrt(bias correction 2) of the n_updates optimiser steps a phase is about to make (adam_step_kernel's own expressions)
__global__ void qf_adam_scalars_kernel(const xrl_adam_state_t* st, int n_updates, float* out) {
#pragma clang fp contract(off)
    const int u = threadIdx.x;
    if (u >= n_updates) return;
    const int step = st->step + 1 + u, sched = st->sched_steps + u;
    const int k = sched < st->total_iters ? sched : st->total_iters;
    const double lr = st->base_lr * (1.0 + (st->end_factor - 1.0) * (double)k / (double)st->total_iters);
    const double b1 = st->beta1, b2 = st->beta2;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    out[2 * u] = (float)(lr / bc1);
    out[2 * u + 1] = (float)sqrt(bc2);
}

// One update of workgroup `wg`: the whole launch of xrl_qmix_fused_update, or update u of a phase (PHASE: xrl_qmix_fused_phase).
// The arguments are read where they lie, in the kernel argument segment (scalar loads, any index): touching a by-value
// parameter with a run-time index would make the compiler copy it to scratch memory first.
template <bool MM, bool ANYACT, bool PHASE>
__device__ __forceinline__ void qf_update_body(const QfArgsK* args, QfPh ph, const int wg, const int u) {
    QfP p = &args->p;
    QfL L = &args->L;
    float* lds = qf_lds;
#define QF_STAMP(i) do { if (p->dbg && blockIdx.x == 0 && threadIdx.x == 0 && (!PHASE || u == ph->n_updates / 2)) p->dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
    QF_STAMP(0);
    const int N = p->N, A = p->A, H = p->H, HH = p->HH, nl = p->n_layers;
    const int b0 = wg * p->items_per_wg;
    const int bw = min(p->items_per_wg, p->B - b0), rows = bw * N, r_base = b0 * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ldq = L->ld[nl];
    QfGlobalOut gslab = (QfGlobalOut)(p->slabs + (size_t)wg * p->slab_stride);

    // ---- 0. ONE burst of global loads: every thread first issues its share of the weight blocks [target mixer | target agent
    //         | eval agent] (float4 chunks of the images) and of the input words (observations, states, per-row scalars,
    //         availability), then the activations' zero padding is written while those loads are in flight (the products read
    //         whole row groups and whole k quads), and only then the loaded values go to LDS: one memory latency in all
    const int ldav = qf_pad4(A), D = p->dims[0], S = p->S;
    const int n_obs = rows * D, n_st = bw * S, n_av = rows * A;
    const int seg1 = n_obs, seg2 = 2 * n_obs, seg3 = seg2 + n_st, seg4 = seg3 + n_st, seg5 = seg4 + 4 * rows, n_in = seg5 + n_av;
    // ring mode: this group's transitions are rows of the replay ring, drawn here (the stream of xrl_sample_replay_indices)
    const bool ring = p->ring_n_envs > 0;
    auto ring_row = [&](int bi) -> size_t {                               // ring row (t * n_envs + env) of transition b0 + bi
        ReplayDraw d;
        d.size_dev = p->size_dev; d.seed = p->draw_seed; d.counter = p->draw_counter + (uint32_t)u; d.counter_dev = p->counter_dev; d.idx_out = nullptr;
        const int64_t fl = replay_draw(d, b0 + bi, p->ring_n_envs, p->ring_n_size);
        const int env = (int)(fl / p->ring_n_size), t = (int)(fl - (int64_t)env * p->ring_n_size);
        if (p->idx_out) p->idx_out[b0 + bi] = fl;                         // (every reader of the row writes the same value)
        return (size_t)t * p->ring_n_envs + env;
    };
    auto in_word = [&](int w, int& dst) -> const float* {               // input word w: where it comes from, where it goes
        if (w < seg2) { const int u = w < seg1 ? w : w - seg1, r = u / D, k = u - r * D;
                        dst = (w < seg1 ? L->x0 : L->x1) + r * L->ld[0] + k;
                        const float* base = w < seg1 ? p->obs : p->obs_next;
                        if (ring) { const int bi = r / N; return base + ring_row(bi) * (size_t)(N * D) + (u - bi * N * D); }
                        return base + (size_t)r_base * D + u; }
        if (w < seg4) { const int u = w < seg3 ? w - seg2 : w - seg3, r = u / S, k = u - r * S;
                        dst = (w < seg3 ? L->s0 : L->s1) + r * L->lds + k;
                        const float* base = w < seg3 ? p->state : p->state_next;
                        if (ring) return base + ring_row(r) * (size_t)S + k;
                        return base + (size_t)b0 * S + u; }
        if (w < seg5) { const int u = w - seg4, f = u / rows, i = u - f * rows;
                        dst = (f == 0 ? L->act_i : f == 1 ? L->rew : f == 2 ? L->term : L->amask) + i;
                        const float* base = f == 0 ? p->actions : f == 1 ? p->rewards : f == 2 ? p->terminals : p->agent_mask;
                        if (ring) { const int bi = i / N; return base + ring_row(bi) * (size_t)N + (i - bi * N); }
                        return base + r_base + i; }
        const int u = w - seg5, r = u / A, k = u - r * A;
        dst = L->avail + r * ldav + k;
        if (!p->avail_next) return nullptr;
        if (ring) { const int bi = r / N; return p->avail_next + ring_row(bi) * (size_t)(N * A) + (u - bi * N * A); }
        return p->avail_next + (size_t)r_base * A + u;
    };
    // (PHASE: the images are rewritten by the other workgroups between two updates -- read past this CU's L1)
    const int img_floats = 4 * (args->agent4 + args->mixer4);
    const __amdgpu_buffer_rsrc_t rs_t = qf_rsrc(p->img_target, PHASE ? img_floats : 0), rs_e = qf_rsrc(p->img_eval, PHASE ? img_floats : 0);
    {
        typedef const __attribute__((address_space(1))) qf_f4* G;
        const int na = args->mixer4, nb = args->agent4, n_w = na + 2 * nb;
        const G sa = (G)(p->img_target + 4 * args->agent4), sb = (G)p->img_target, sc = (G)p->img_eval;
        qf_f4 wv[8];
        float iv[2];
        int idst[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = tid + j * QF_THREADS;
            wv[j] = 0.f;
            if constexpr (PHASE) {
                if (q < na) wv[j] = qf_ld4_dev(rs_t, args->agent4 + q);
                else if (q < na + nb) wv[j] = qf_ld4_dev(rs_t, q - na);
                else if (q < n_w) wv[j] = qf_ld4_dev(rs_e, q - na - nb);
            } else {
            if (q < na) wv[j] = sa[q];
            else if (q < na + nb) wv[j] = sb[q - na];
            else if (q < n_w) wv[j] = sc[q - na - nb];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int w = tid + j * QF_THREADS;
            idst[j] = -1; iv[j] = 1.f;
            if (w < n_in) { const float* src = in_word(w, idst[j]); if (src) iv[j] = *src; }
        }
        for (int i = tid; i < L->clear_end; i += QF_THREADS) lds[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) if (idst[j] >= 0) lds[idst[j]] = iv[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = tid + j * QF_THREADS;
            if (q < na) *reinterpret_cast<qf_f4*>(lds + L->mix + 4 * q) = wv[j];
            else if (q < na + nb) *reinterpret_cast<qf_f4*>(lds + L->ag_t + 4 * (q - na)) = wv[j];
            else if (q < n_w) *reinterpret_cast<qf_f4*>(lds + L->ag_e + 4 * (q - na - nb)) = wv[j];
        }
        for (int w = tid + 2 * QF_THREADS; w < n_in; w += QF_THREADS) {       // (larger groups: the rest, one more latency)
            int dst; const float* src = in_word(w, dst);
            lds[dst] = src ? *src : 1.f;
        }
        for (int q = tid + 8 * QF_THREADS; q < n_w; q += QF_THREADS) {        // (larger networks: likewise)
            if (q < na) *reinterpret_cast<qf_f4*>(lds + L->mix + 4 * q) = PHASE ? qf_ld4_dev(rs_t, args->agent4 + q) : sa[q];
            else if (q < na + nb) *reinterpret_cast<qf_f4*>(lds + L->ag_t + 4 * (q - na)) = PHASE ? qf_ld4_dev(rs_t, q - na) : sb[q - na];
            else *reinterpret_cast<qf_f4*>(lds + L->ag_e + 4 * (q - na - nb)) = PHASE ? qf_ld4_dev(rs_e, q - na - nb) : sc[q - na - nb];
        }
    }
    __syncthreads();

    QF_STAMP(1);
    // ---- 1..4. forward.  Between two barriers run, side by side on disjoint thread ranges: layer l of the three agent
    //            passes (target(next), eval(next) for the double-Q argmax, eval(obs) kept for the backward pass) and one
    //            step of the mixer sequence: target hyper layer A, target hyper layer B, eval mixer weights over the target
    //            mixer's LDS space, eval hyper layer A, eval hyper layer B
    // products: VALU loops, or -- enough rows to fill a tile -- matrix-core tiles dealt round-robin over the waves (tb: tiles handed out
    // in the current phase)
    constexpr bool mm = MM;                                                                  // (the launcher: qf_mfma_products)
    int tb = 0;
    auto FWD = [&](int W_, int ldw_, int b_, int K_, int Nn_, int in_, int ldi_, int rows_, int out_, int ldo_, int act_, int tid0_) {
        if constexpr (mm) {
            const int t = qf_tiles(rows_, Nn_), t0 = 64 * (tb & 15);
            tb += t;
            if (qf_mm_wave_in(t, t0)) qf_mm_fwd_fn(W_, ldw_, b_, K_, Nn_, in_, ldi_, rows_, out_, ldo_, act_, t0);
        } else qf_lin_fwd<ANYACT>(W_, ldw_, b_, K_, Nn_, in_, ldi_, rows_, out_, ldo_, act_, tid0_);
    };
    auto BWDD = [&](int W_, int ldw_, int K_, int Nn_, int dz_, int ldz_, int rows_, int dx_, int ldx_, int y_, int ldy_, int act_, int tid0_) {
        if constexpr (mm) {
            const int t = qf_tiles(rows_, K_), t0 = 64 * (tb & 15);
            tb += t;
            if (qf_mm_wave_in(t, t0)) qf_mm_bwd_data_fn(W_, ldw_, K_, Nn_, dz_, ldz_, rows_, dx_, ldx_, y_, ldy_, act_, t0);
        } else qf_lin_bwd_data(W_, ldw_, K_, Nn_, dz_, ldz_, rows_, dx_, ldx_, y_, ldy_, act_, tid0_);
    };
    auto BWDW = [&](QfGlobalOut dW_, QfGlobalOut db_, int K_, int Nn_, int dz_, int ldz_, int in_, int ldi_, int rows_, int tid0_) {
        if constexpr (mm) {
            const int t = qf_tiles(Nn_, K_), t0 = 64 * (tb & 15);
            tb += t;
            if (qf_mm_wave_in(t, t0)) qf_mm_bwd_weight_fn(dW_, db_, K_, Nn_, dz_, ldz_, in_, ldi_, rows_, t0);
        } else qf_lin_bwd_weight(dW_, db_, K_, Nn_, dz_, ldz_, in_, ldi_, rows_, tid0_);
    };
    // hyper-networks of the mixer staged in LDS (q_mix_head.py:50-64), one layer per call (no barrier inside):
    // A: hid = [relu(W_f s + b_f) (3 HH) | W_b1 s + b_b1 (H)];  B: raw = [w1 | w2 | b2] from hid
    auto HYPER = [&](int layer, int s_, int hid_, int raw_, int tid0_) {
        if (layer == 0) {
            FWD(L->mw[0], L->mldw[0], L->mb[0], p->S, 3 * HH, s_, L->lds, bw, hid_, L->ldh, XRL_ACT_RELU, tid0_);
            FWD(L->mw[1], L->mldw[1], L->mb[1], p->S, H, s_, L->lds, bw, hid_ + 3 * HH, L->ldh, XRL_ACT_NONE, tid0_ + 128);
        } else {
            FWD(L->mw[2], L->mldw[2], L->mb[2], HH, N * H, hid_, L->ldh, bw, raw_, L->ldr, XRL_ACT_NONE, tid0_);
            FWD(L->mw[3], L->mldw[3], L->mb[3], HH, H, hid_ + HH, L->ldh, bw, raw_ + N * H, L->ldr, XRL_ACT_NONE, tid0_ + 128);
            FWD(L->mw[4], L->mldw[4], L->mb[4], HH, 1, hid_ + 2 * HH, L->ldh, bw, raw_ + N * H + H, L->ldr, XRL_ACT_NONE, tid0_ + 192);
        }
    };
    int mix_step = 0;
    for (int l = 0; l < nl || mix_step < 5; ++l) {
        tb = 0;
        if (l < nl) {
            const bool last = l == nl - 1;
            const int act = last ? XRL_ACT_NONE : p->act, K = p->dims[l], Nn = p->dims[l + 1], ldi = L->ld[l], ldo = L->ld[l + 1];
            const int in_t = l == 0 ? L->x1 : ((l & 1) ? L->t0 : L->t1), out_t = last ? L->qnt : (((l + 1) & 1) ? L->t0 : L->t1);
            const int in_n = l == 0 ? L->x1 : ((l & 1) ? L->u0 : L->u1), out_n = last ? L->qne : (((l + 1) & 1) ? L->u0 : L->u1);
            const int in_e = l == 0 ? L->x0 : L->h[l], out_e = last ? L->q : L->h[l + 1];
            FWD(L->wt[l], L->ldw[l], L->bt[l], K, Nn, in_t, ldi, rows, out_t, ldo, act, 0);
            if (p->double_q) FWD(L->we[l], L->ldw[l], L->be[l], K, Nn, in_n, ldi, rows, out_n, ldo, act, 256);
            FWD(L->we[l], L->ldw[l], L->be[l], K, Nn, in_e, ldi, rows, out_e, ldo, act, 512);
        }
        switch (mix_step) {
            case 0: HYPER(0, L->s1, L->hid_t, L->raw_t, 768); break;
            case 1: HYPER(1, L->s1, L->hid_t, L->raw_t, 768); break;
            case 2: if constexpr (PHASE) qf_copy_dev(p->img_eval + 4 * args->agent4, 4 * args->mixer4, L->mix, args->mixer4);
                    else qf_copy((QfGlobalIn)(p->img_eval + 4 * args->agent4), L->mix, args->mixer4);
                    break;
            case 3: HYPER(0, L->s0, L->hid_e, L->raw_e, 768); break;
            case 4: HYPER(1, L->s0, L->hid_e, L->raw_e, 768); break;
            default: break;
        }
        ++mix_step;
        if (p->dbg && blockIdx.x == 0 && u == 0 && (threadIdx.x & 63) == 0 && l < 6) p->dbg[32 + 16 * l + (threadIdx.x >> 6)] = (long long)__builtin_readcyclecounter();   // (diagnostics: [128])
        __syncthreads();
        if (l < 6) QF_STAMP(16 + l);
    }
    QF_STAMP(4);

    // ---- 5. one wavefront per transition: taken / target Q, mixing, TD error, backward to d Q_eval and d(hyper outputs)
    //         (the arithmetic of xrl_qmix_mix_td, statement by statement; lane n < N = agent n, lane h < H = hidden unit h)
    float* dq = lds + L->t0;                                             // [rows][ldq]: d loss / d Q_eval(obs)
    for (int i = tid; i < L->rows_pad * ldq; i += QF_THREADS) dq[i] = 0.f;
    __syncthreads();
    for (int bi = wave; bi < bw; bi += QF_THREADS / 64) {
        const int b = b0 + bi;
        float qe = 0.f, qn = 0.f, mask = 0.f;
        int a_taken = 0;
        if (lane < N) {
            const int lr = bi * N + lane;
            mask = lds[L->amask + lr];
            a_taken = (int)lds[L->act_i + lr];
            qe = lds[L->q + lr * ldq + a_taken] * mask;                                   // qmix_learner.py:48-50,60
            const float* qt = lds + L->qnt + lr * ldq;
            const float* av = lds + L->avail + lr * ldav;
            if (p->double_q) {                                                            // :52-55
                const float* qs = lds + L->qne + lr * ldq;
                int best = 0; float bv = (av && av[0] == 0.f) ? -1e10f : qs[0];
                for (int j = 1; j < A; ++j) {
                    const float v = (av && av[j] == 0.f) ? -1e10f : qs[j];               // value_factorization.py:87-90
                    if (v > bv) { bv = v; best = j; }
                }
                qn = (av && av[best] == 0.f) ? -1e10f : qt[best];
            } else {                                                                     // :57-58
                qn = (av && av[0] == 0.f) ? -1e10f : qt[0];
                for (int j = 1; j < A; ++j) qn = fmaxf(qn, (av && av[j] == 0.f) ? -1e10f : qt[j]);
            }
            qn *= mask;                                                                  // :61
        }
        const float* e_raw = lds + L->raw_e + bi * L->ldr;
        const float* t_raw = lds + L->raw_t + bi * L->ldr;
        float pre_e = 0.f, pre_t = 0.f, w2e = 0.f, w2t = 0.f;
        if (lane < H) { pre_e = lds[L->hid_e + bi * L->ldh + 3 * HH + lane]; pre_t = lds[L->hid_t + bi * L->ldh + 3 * HH + lane]; }
        for (int n = 0; n < N; ++n) {
            const float qen = __shfl(qe, n, 64), qnn = __shfl(qn, n, 64);
            if (lane < H) {
                pre_e += qen * fabsf(e_raw[n * H + lane]);                               // bmm(agent_qs, |w1|) + b1
                pre_t += qnn * fabsf(t_raw[n * H + lane]);
            }
        }
        float hid_e = 0.f, hid_t = 0.f;
        if (lane < H) {
            hid_e = qf_elu(pre_e); hid_t = qf_elu(pre_t);
            w2e = fabsf(e_raw[N * H + lane]); w2t = fabsf(t_raw[N * H + lane]);
        }
        const float q_tot_e = wave_sum(hid_e * w2e) + e_raw[N * H + H];                  // bmm(hidden, |w2|) + b2
        const float q_tot_n = wave_sum(hid_t * w2t) + t_raw[N * H + H];
        float r = 0.f, dn = 1.f;
        if (lane < N) { r = lds[L->rew + bi * N + lane]; dn = lds[L->term + bi * N + lane] != 0.f ? 1.f : 0.f; }
        const float r_tot = wave_sum(r) / (float)N;                                      // :34
        float all_d = dn;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) all_d = fminf(all_d, __shfl_xor(all_d, off, 64));   // :35
        const float y = r_tot + (1.f - all_d) * p->gamma * q_tot_n;                       // :78
        const float td = q_tot_e - y;
        const float dq_tot = 2.f * td / (float)p->B;                                      // d mean(td^2) / d q_tot_eval (:86)
        float* d_raw = lds + L->d_raw + bi * L->ldr;
        float d_pre = 0.f;
        if (lane < H) {
            const float w2_raw = e_raw[N * H + lane];
            const float sgn2 = (w2_raw > 0.f) - (w2_raw < 0.f);
            d_raw[N * H + lane] = dq_tot * hid_e * sgn2;                                 // through abs()
            d_pre = dq_tot * w2e * (pre_e > 0.f ? 1.f : expf(pre_e));                    // ELU'
            lds[L->d_hid + bi * L->ldh + 3 * HH + lane] = d_pre;                           // = d hyper_b_1 output
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
            const float dqe = wave_sum(contrib);                                         // d loss / d (masked q_eval_n)
            if (lane == n) dq[(bi * N + n) * ldq + a_taken] = dqe * mask;
        }
        if (lane == 0) {
            double* o = (PHASE ? ph->phase_partials + (size_t)u * p->B * 8 : p->partials) + (size_t)b * 8;
            o[0] = (double)td * td; o[1] = q_tot_e; o[2] = 0.0;
            for (int j = 3; j < 8; ++j) o[j] = 0.0;
            if (p->diag) { p->diag[b] = q_tot_e; p->diag[p->B + b] = q_tot_n; p->diag[2 * (size_t)p->B + b] = y; }
        }
    }
    __syncthreads();

    QF_STAMP(6);
    // ---- 6..7. backward, again side by side between barriers: the agent network from d Q_eval(obs) (weight gradient of
    //            layer l, data gradient into layer l - 1) and the eval hyper-networks (layer B's weight and data gradients,
    //            then layer A's weight gradients); the target networks have no gradient
    {
        const int d_hid = L->d_hid, hid = L->hid_e, d_raw = L->d_raw;
        int dz = L->t0, ldz = ldq, hyper_step = 0;                        // dz: d (pre-activation of layer l's output)
        for (int l = nl - 1; l >= 0 || hyper_step < 2; --l) {
            tb = 0;
            if (hyper_step == 0) {
                BWDW(gslab + p->mix_off[XRL_QF_W1_W], gslab + p->mix_off[XRL_QF_W1_B], HH, N * H, d_raw, L->ldr, hid, L->ldh, bw, 0);
                BWDW(gslab + p->mix_off[XRL_QF_W2_W], gslab + p->mix_off[XRL_QF_W2_B], HH, H, d_raw + N * H, L->ldr, hid + HH, L->ldh, bw, 768);
                BWDW(gslab + p->mix_off[XRL_QF_B2_W], gslab + p->mix_off[XRL_QF_B2_B], HH, 1, d_raw + N * H + H, L->ldr, hid + 2 * HH, L->ldh, bw, 960);
                BWDD(L->mw[2], L->mldw[2], HH, N * H, d_raw, L->ldr, bw, d_hid, L->ldh, hid, L->ldh, XRL_ACT_RELU, 256);
                BWDD(L->mw[3], L->mldw[3], HH, H, d_raw + N * H, L->ldr, bw, d_hid + HH, L->ldh, hid + HH, L->ldh, XRL_ACT_RELU, 384);
                BWDD(L->mw[4], L->mldw[4], HH, 1, d_raw + N * H + H, L->ldr, bw, d_hid + 2 * HH, L->ldh, hid + 2 * HH, L->ldh, XRL_ACT_RELU, 448);
            } else if (hyper_step == 1) {
                BWDW(gslab + p->mix_off[XRL_QF_FIRST_W], gslab + p->mix_off[XRL_QF_FIRST_B], p->S, 3 * HH, d_hid, L->ldh, L->s0, L->lds, bw, 0);
                BWDW(gslab + p->mix_off[XRL_QF_B1_W], gslab + p->mix_off[XRL_QF_B1_B], p->S, H, d_hid + 3 * HH, L->ldh, L->s0, L->lds, bw, 384);
            }
            ++hyper_step;
            if (l >= 0) {
                const int in = l == 0 ? L->x0 : L->h[l], ldi = L->ld[l];
                BWDW(gslab + p->w_off[l], gslab + p->b_off[l], p->dims[l], p->dims[l + 1], dz, ldz, in, ldi, rows, 512);
                if (l > 0) {
                    const int dx = (dz == L->t0) ? L->t1 : L->t0;
                    // t0 / t1 rows are ldmax wide; the first dz (the Q head's) uses ldq, every later one ld[l]
                    BWDD(L->we[l], L->ldw[l], p->dims[l], p->dims[l + 1], dz, ldz, rows, dx, L->ld[l], in, ldi, p->act, 640);
                    dz = dx; ldz = L->ld[l];
                }
            }
            __syncthreads();
        }
    }
    QF_STAMP(8);
}





// PHASE (xrl_qmix_fused_phase): the workgroups stay for all updates of a phase and do the optimiser steps between them
template <bool MM, bool ANYACT, bool PHASE>
__global__ void __launch_bounds__(QF_THREADS) qmix_fused_kernel(typename qf_kernel_arg<PHASE>::type by_value_unused) {
    const QfArgsK* args = (const QfArgsK*)__builtin_amdgcn_kernarg_segment_ptr();      // (QfArgs is the first member of QfArgsPh)
    if constexpr (!PHASE) {
        qf_update_body<MM, ANYACT, false>(args, nullptr, (int)blockIdx.x, 0);
    } else {
        if (blockIdx.x & 7) return;                                      // one XCD's share of the grid (rollout_actor.hip)
        const int wg = (int)(blockIdx.x >> 3), n_wg = (int)(gridDim.x >> 3);
        QfPh ph = (QfPh)((const __attribute__((address_space(4))) char*)args + offsetof(QfArgsPh, ph));
        // (read before this workgroup's first arrival: workgroup 0 advances the state behind every second meeting)
        const unsigned step0 = (unsigned)__hip_atomic_load(&ph->state->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned sched0 = (unsigned)__hip_atomic_load(&ph->state->sched_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned seen = atomicOr(ph->sync + QFS_MASK, 1u << (xcc & 0xf));
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
        }
        bool multi = true;                                               // placement unknown until the first meeting: fences
        const int n_upd = ph->n_updates;
        for (int u = 0; u < n_upd; ++u) {
            // (the argument pointer is laundered per update: with the body inlined into this loop everything it reads from the argument
            //  segment was hoisted out of the loop and kept live across it -- 220 spilled SGPRs, 127 spilled VGPRs; as a real call the body
            //  saved ~50 callee-saved registers through scratch per update)
            const QfArgsK* a = args;
            asm volatile("" : "+s"(a));
            QfPh ph_u = (QfPh)((const __attribute__((address_space(4))) char*)a + offsetof(QfArgsPh, ph));
            qf_update_body<false, ANYACT, true>(a, ph_u, wg, u);
            multi = qf_phase_step(ph_u, &a->p, &a->L, wg, n_wg, u, n_upd, step0, sched0, multi);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// One ACTING step of the recurrent agents (value_factorization.py:66-92 -> Basic_RNN, rnn.py:52-77: mlp blocks -> nn.GRU
// cell -> Q head) for R rows as ONE launch: the layered path is four GEMM launches of ~7 us on 192 rows plus the recurrence
// launch.  Same scheme as the update above: rows_per_wg rows per workgroup, the whole weight image staged in LDS in one
// burst, layer after layer from LDS, W_ih x and W_hh h side by side; the cell arithmetic is csrc/gru.hip's.
struct QaLds {
    int x, hin, a0, a1, gi, gh, hnew, q, reset, clear_end, img, total;
    int ldx, ldh, lda, ldg, ldq, rows_pad;
    int w[XRL_QA_MAX_LAYERS], b[XRL_QA_MAX_LAYERS], ldw[XRL_QA_MAX_LAYERS];    // image offsets (from the image base)
    int image_floats;
};

// the shapes marl_act_rows_kernel takes: one row per workgroup, every layer <= 64 wide, H = 64 (or 0), O <= 64; lds_staged forces
// the other kernel (tests: the two must agree bit for bit)
__host__ __device__ inline bool qa_rows_form(const xrl_marl_act_gru_t& p) {
    if (p.lds_staged || p.rows_per_wg != 1 || p.O > 64 || !(p.H == 0 || p.H == 64) || p.n_pre > 3 || p.n_post > 3) return false;
    for (int i = 0; i < p.n_pre; ++i) if (p.pre[i] > 64) return false;
    for (int i = 0; i < p.n_post; ++i) if (p.post[i] > 64) return false;
    return true;
}

__host__ __device__ inline void qa_layers(const xrl_marl_act_gru_t& p, int* K, int* Nn, int& n_layers) {
    // layer list in image order: pre[0..n_pre), ih, hh, post[0..n_post)
    int l = 0, feat = p.O;
    for (int i = 0; i < p.n_pre; ++i) { K[l] = feat; Nn[l] = p.pre[i]; feat = p.pre[i]; ++l; }
    if (p.H > 0) {                                       // H == 0: feed-forward agents, no recurrent layer
        K[l] = feat; Nn[l] = 3 * p.H; ++l;
        K[l] = p.H; Nn[l] = 3 * p.H; ++l;
        feat = p.H;
    }
    for (int i = 0; i < p.n_post; ++i) { K[l] = feat; Nn[l] = p.post[i]; feat = p.post[i]; ++l; }
    n_layers = l;
}

__host__ __device__ inline QaLds qa_layout(const xrl_marl_act_gru_t& p) {
    QaLds L;
    int K[XRL_QA_MAX_LAYERS], Nn[XRL_QA_MAX_LAYERS], nl;
    qa_layers(p, K, Nn, nl);
    int off = 0, amax = p.H;
    for (int i = 0; i < p.n_pre; ++i) if (p.pre[i] > amax) amax = p.pre[i];
    for (int i = 0; i + 1 < p.n_post; ++i) if (p.post[i] > amax) amax = p.post[i];
    L.rows_pad = (p.rows_per_wg + QF_PAD - 1) / QF_PAD * QF_PAD;
    L.ldx = qf_pad4(p.O); L.ldh = qf_pad4(p.H); L.lda = qf_pad4(amax); L.ldg = qf_pad4(3 * p.H); L.ldq = qf_pad4(p.post[p.n_post - 1]);
    L.x = off; off += L.rows_pad * L.ldx;
    L.hin = off; off += L.rows_pad * L.ldh;
    L.a0 = off; off += L.rows_pad * L.lda;
    L.a1 = off; off += L.rows_pad * L.lda;
    L.gi = off; off += L.rows_pad * L.ldg;
    L.gh = off; off += L.rows_pad * L.ldg;
    L.hnew = off; off += L.rows_pad * L.ldh;
    L.q = off; off += L.rows_pad * L.ldq;
    L.reset = off; off += L.rows_pad;
    L.clear_end = off;
    L.img = off;
    int io = 0;
    for (int l = 0; l < XRL_QA_MAX_LAYERS; ++l) { L.w[l] = L.b[l] = L.ldw[l] = 0; }
    if (qa_rows_form(p)) {        // marl_act_rows_kernel: k-quads interleaved over the outputs (ldw = outputs padded to a wave)
        for (int l = 0; l < nl; ++l) { L.ldw[l] = (Nn[l] + 63) / 64 * 64; L.w[l] = io; io += ((K[l] + 3) / 4) * L.ldw[l] * 4; }
    } else {
        for (int l = 0; l < nl; ++l) { L.ldw[l] = qf_pad4(K[l]) + 4; L.w[l] = io; io += Nn[l] * L.ldw[l]; }
    }
    for (int l = 0; l < nl; ++l) { L.b[l] = io; io += qf_pad4(Nn[l]); }
    L.image_floats = io;
    L.total = off + io;
    return L;
}

struct QaArgs { xrl_marl_act_gru_t p; QaLds L; };
typedef const __attribute__((address_space(4))) QaArgs QaArgsK;

__device__ __forceinline__ float qa_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float qa_tanh(float x) { return __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)), 1.f); }
// the GRU cell of both acting kernels, every fused multiply-add spelled out (under `contract(fast)` the compiler is free to fuse or not,
// and did it differently at different sites: the two kernels must round alike)
__device__ __forceinline__ float qa_cell(float gi_r, float gi_z, float gi_n, float gh_r, float gh_z, float gh_n, float h) {
    const float rg = qa_sigmoid(gi_r + gh_r);
    const float z = qa_sigmoid(gi_z + gh_z);
    const float n = qa_tanh(__builtin_fmaf(rg, gh_n, gi_n));
    return __builtin_fmaf(h - n, z, n);
}

// tools/probe_act_gru.py: per-workgroup clock stamps of the acting launch's phases (xrl_debug_act_gru_stamps; NULL = off)
__device__ long long* g_qa_dbg = nullptr;
#define QSTAMP(k) do { if (qdbg && threadIdx.x == 0) qdbg[16 * blockIdx.x + (k)] = (k) == 0 || (k) == 15 ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_amdgcn_s_memtime(); } while (0)

template <bool ANYACT>
__global__ void __launch_bounds__(QF_THREADS) marl_act_gru_kernel(QaArgs by_value_unused) {
    const QaArgsK* args = (const QaArgsK*)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) xrl_marl_act_gru_t* p = &args->p;
    const __attribute__((address_space(4))) QaLds* L = &args->L;
    float* lds = qf_lds;
    const int tid = threadIdx.x, H = p->H, O = p->O;
    long long* const qdbg = g_qa_dbg;
    QSTAMP(0); QSTAMP(1);
    const int r0 = blockIdx.x * p->rows_per_wg, rows = min(p->rows_per_wg, p->R - r0);
    // ---- one burst: weight image, observations, previous hidden state, reset flags
    {
        typedef const __attribute__((address_space(1))) qf_f4* G;
        const G src = (G)p->image;
        const int n4 = L->image_floats >> 2, n_x = rows * O, n_h = rows * H, n_in = n_x + n_h + (H > 0 ? rows : 0);
        qf_f4 wv[10];
        float iv[2];
        int idst[2];
#pragma unroll
        for (int j = 0; j < 10; ++j) { const int q = tid + j * QF_THREADS; wv[j] = 0.f; if (q < n4) wv[j] = src[q]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int w = tid + j * QF_THREADS;
            idst[j] = -1; iv[j] = 0.f;
            if (w < n_x) { const int r = w / O, k = w - r * O; idst[j] = L->x + r * L->ldx + k; iv[j] = p->obs[(size_t)r0 * O + w]; }
            else if (w < n_x + n_h) { const int u = w - n_x, r = u / H, k = u - r * H; idst[j] = L->hin + r * L->ldh + k; iv[j] = p->h[(size_t)r0 * H + u]; }
            else if (w < n_in) { const int r = w - n_x - n_h; idst[j] = L->reset + r; iv[j] = p->reset ? p->reset[r0 + r] : 0.f; }
        }
        for (int i = tid; i < L->clear_end; i += QF_THREADS) lds[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) if (idst[j] >= 0) lds[idst[j]] = iv[j];
#pragma unroll
        for (int j = 0; j < 10; ++j) { const int q = tid + j * QF_THREADS; if (q < n4) *reinterpret_cast<qf_f4*>(lds + L->img + 4 * q) = wv[j]; }
        for (int q = tid + 10 * QF_THREADS; q < n4; q += QF_THREADS) *reinterpret_cast<qf_f4*>(lds + L->img + 4 * q) = src[q];
        for (int w = tid + 2 * QF_THREADS; w < n_in; w += QF_THREADS) {
            if (w < n_x) { const int r = w / O, k = w - r * O; lds[L->x + r * L->ldx + k] = p->obs[(size_t)r0 * O + w]; }
            else if (w < n_x + n_h) { const int u = w - n_x, r = u / H, k = u - r * H; lds[L->hin + r * L->ldh + k] = p->h[(size_t)r0 * H + u]; }
            else { const int r = w - n_x - n_h; lds[L->reset + r] = p->reset ? p->reset[r0 + r] : 0.f; }
        }
    }
    __syncthreads();
    QSTAMP(2);
    for (int i = tid; i < rows * H; i += QF_THREADS) {                    // init_rnn_states_item (rnn.py:86-92): zero state
        const int r = i / H, k = i - r * H;
        if (lds[L->reset + r] != 0.f) lds[L->hin + r * L->ldh + k] = 0.f;
    }
    // ---- layers below the recurrence
    int l = 0, in = L->x, ldi = L->ldx, feat = O;
    for (int i = 0; i < p->n_pre; ++i, ++l) {
        const int out = (i & 1) ? L->a1 : L->a0;
        qf_lin_fwd<ANYACT>(L->img + L->w[l], L->ldw[l], L->img + L->b[l], feat, p->pre[i], in, ldi, rows, out, L->lda, p->act, 0);
        __syncthreads();
        in = out; ldi = L->lda; feat = p->pre[i];
    }
    QSTAMP(3);
    if (H > 0) {
        // ---- gi = W_ih x + b_ih,  gh = W_hh h + b_hh  (side by side), then the cell (csrc/gru.hip's arithmetic)
        qf_lin_fwd<ANYACT>(L->img + L->w[l], L->ldw[l], L->img + L->b[l], feat, 3 * H, in, ldi, rows, L->gi, L->ldg, XRL_ACT_NONE, 0);
        qf_lin_fwd<ANYACT>(L->img + L->w[l + 1], L->ldw[l + 1], L->img + L->b[l + 1], H, 3 * H, L->hin, L->ldh, rows, L->gh, L->ldg, XRL_ACT_NONE, 512);
        l += 2;
        __syncthreads();
        QSTAMP(4);
        for (int i = tid; i < rows * H; i += QF_THREADS) {
            const int r = i / H, j = i - r * H;
            const float* gi = lds + L->gi + r * L->ldg;
            const float* gh = lds + L->gh + r * L->ldg;
            const float h = lds[L->hin + r * L->ldh + j];
            const float hn = qa_cell(gi[j], gi[H + j], gi[2 * H + j], gh[j], gh[H + j], gh[2 * H + j], h);
            lds[L->hnew + r * L->ldh + j] = hn;
            p->h[(size_t)(r0 + r) * H + j] = hn;                               // the state carried to the next step
        }
        __syncthreads();
        QSTAMP(5);
        in = L->hnew; ldi = L->ldh; feat = H;
    }
    // ---- Q head
    for (int i = 0; i < p->n_post; ++i, ++l) {
        const bool last = i == p->n_post - 1;
        const int out = last ? L->q : ((i & 1) ? L->a1 : L->a0), ldo = last ? L->ldq : L->lda;
        qf_lin_fwd<ANYACT>(L->img + L->w[l], L->ldw[l], L->img + L->b[l], feat, p->post[i], in, ldi, rows, out, ldo, last ? XRL_ACT_NONE : p->act, 0);
        __syncthreads();
        in = out; ldi = ldo; feat = p->post[i];
    }
    QSTAMP(6);
    const int A = p->post[p->n_post - 1];
    for (int i = tid; i < rows * A; i += QF_THREADS) {
        const int r = i / A, k = i - r * A;
        p->q[(size_t)(r0 + r) * p->ldq + k] = lds[L->q + r * L->ldq + k];
    }
    // ---- optional: the step's action selection for these rows (xrl_marl_select_actions' arithmetic and Philox keys)
    if (p->action && tid < rows) {
        const int r = r0 + tid;
        const uint32_t step = p->step + (p->step_dev ? *p->step_dev : 0u);
        const int a = marl_select_row(lds + L->q + tid * L->ldq, p->avail ? p->avail + (size_t)r * A : nullptr, A, p->seed, step, r,
                                      p->eps_dev ? *p->eps_dev : p->eps, nullptr, nullptr);
        p->action[r] = a;
        if (p->action_f) p->action_f[r] = (float)a;
    }
    QSTAMP(7); QSTAMP(15);
}
// ---- the same step with ONE THREAD PER OUTPUT and the weights in registers (round 6; tools/probe_act_gru.py, profiles/r06_k_act_gru.json).
// marl_act_gru_kernel above stages the whole weight image (108 KB for 3m.yaml's agents) in LDS before anything starts and its products
// read weights AND inputs from LDS: stamps of a 6-row workgroup -- image burst 6.3 k cycles, fc 3.0 k, gate products 10.9 k (LDS return
// bandwidth: five 16-byte reads per 16 fma), cell 0.9 k, Q head 4.6 k (54 work items walking K = 64 one after the other behind a call
// each), stores + selection 4.1 k (dependent loads of mask / epsilon / step).  Here a workgroup takes ONE row and 12 waves:
//   waves 0..5   thread n < 3 H: row n of W_ih, thread 3 H + n: row n of W_hh        (H == 64; idle for feed-forward agents)
//   waves 6..8   pre layer 0..2, lane = output
//   waves 9..11  post layer 0..2, lane = output
// every thread pulls its weight row (<= 64 floats) and bias from the image in global memory straight into registers while observations /
// hidden state / mask go to LDS; the image is INTERLEAVED for that (xrl_qa_image_t.interleaved: the four weights k = 4 i .. 4 i + 3 of
// output n at w[l] + (i * ldw[l] + n) * 4, ldw = outputs padded to 64: a wave's load is one contiguous KB -- with row-major rows every
// 16-byte load of a wave touched 64 cache lines and the burst alone took 8-9 k cycles); the inputs of a product are broadcast LDS reads;
// W_hh h does not wait for the layers below.  Every output is the same fma chain over k = 0, 1, ... as qf_lin_fwd_t's: results are
// bit-identical to the kernel above (tests/test_gpu_offpolicy_agents.py).  One row per workgroup: with two the compiler keeps both rows'
// 16 input quads live next to the 64 weight registers and spills (85 registers at two rows, 581 at four; 18.7 / 48 us per launch).
constexpr int QR_THREADS = 768, QR_ROWS = 1, QR_LD = 68, QR_LDG = 196;
constexpr int QR_X = 0, QR_H = QR_X + QR_ROWS * QR_LD, QR_A0 = QR_H + QR_ROWS * QR_LD, QR_A1 = QR_A0 + QR_ROWS * QR_LD,
              QR_HN = QR_A1 + QR_ROWS * QR_LD, QR_Q = QR_HN + QR_ROWS * QR_LD, QR_GI = QR_Q + QR_ROWS * QR_LD,
              QR_GH = QR_GI + QR_ROWS * QR_LDG, QR_AV = QR_GH + QR_ROWS * QR_LDG, QR_RNG = QR_AV + QR_ROWS * 64, QR_TOTAL = QR_RNG + 4;

// one output for every row of the workgroup: acc_r = sum_k x_r[k] w[k] in qf_lin_fwd_t's order, + bias (x: LDS offset, rows of QR_LD).
// All 16 k-quads of all RPW rows, no predicates: weights beyond K are zero, inputs beyond K / rows beyond the batch are zero-filled LDS
// (adding x * 0 leaves the sum as it is), so the 16 reads of a row are in flight together -- with `if (i < K4)` / `if (j < rows)` around
// them every read was a branch target of its own and its latency stood in the chain: 3.7 k cycles for the 30-wide fc product of one row
template <int RPW>
__device__ __forceinline__ void qr_product(const float* lds, int in, const qf_f4 (&w)[16], float bias, float (&v)[RPW]) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        float4 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = *reinterpret_cast<const float4*>(lds + in + j * QR_LD + 4 * i);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {                         // (spelled out: left to `contract(fast)` one of the five call sites became pk_mul + add)
            acc = __builtin_fmaf(x[i].x, w[i].x, acc); acc = __builtin_fmaf(x[i].y, w[i].y, acc);
            acc = __builtin_fmaf(x[i].z, w[i].z, acc); acc = __builtin_fmaf(x[i].w, w[i].w, acc);
        }
        v[j] = acc + bias;
    }
}

template <bool ANYACT>
__global__ void __launch_bounds__(QR_THREADS) marl_act_rows_kernel(QaArgs by_value_unused) {
    const QaArgsK* args = (const QaArgsK*)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) xrl_marl_act_gru_t* p = &args->p;
    const __attribute__((address_space(4))) QaLds* L = &args->L;
    __shared__ __attribute__((aligned(16))) float lds[QR_TOTAL];
    typedef const __attribute__((address_space(1))) qf_f4* G4;
    typedef const __attribute__((address_space(1))) float* G1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, H = p->H, O = p->O;
    constexpr int RPW = QR_ROWS;
    const int r0 = blockIdx.x, rows = 1;
    const int n_pre = p->n_pre, n_post = p->n_post, A = p->post[n_post - 1], act = p->act;
    long long* const qdbg = g_qa_dbg;
    QSTAMP(0); QSTAMP(1);
    // ---- this thread's role: layer l of the image, output n, K inputs
    int l = -1, n = 0, K = 0;
    {
        const int feat_pre = n_pre > 0 ? p->pre[n_pre - 1] : O;
        if (wave < 6) {
            if (H > 0 && tid < 6 * H) { l = n_pre + (tid >= 3 * H ? 1 : 0); n = tid >= 3 * H ? tid - 3 * H : tid; K = tid >= 3 * H ? H : feat_pre; }
        } else if (wave < 9) {
            const int i = wave - 6;
            if (i < n_pre && lane < p->pre[i]) { l = i; n = lane; K = i == 0 ? O : p->pre[i - 1]; }
        } else {
            const int i = wave - 9;
            if (i < n_post && lane < p->post[i]) { l = n_pre + (H > 0 ? 2 : 0) + i; n = lane; K = i == 0 ? (H > 0 ? H : feat_pre) : p->post[i - 1]; }
        }
    }
    const int K4 = (K + 3) >> 2;
    // ---- weights and bias -> registers (loads in flight while the inputs are staged)
    qf_f4 w[16];
    float bias = 0.f;
    {
        const int lw = l >= 0 ? l : 0;
        const G4 wp = (G4)(p->image + L->w[lw]) + n;
        const int np = L->ldw[lw];
#pragma unroll
        for (int i = 0; i < 16; ++i) { w[i] = 0.f; if (l >= 0 && i < K4) w[i] = wp[i * np]; }
        if (l >= 0) bias = ((G1)p->image)[L->b[lw] + n];
    }
    // ---- inputs: observations (pad zero), hidden state (zero where the row starts an episode, rnn.py:86-92), action mask
    {
        float xv = 0.f, hv = 0.f, rv = 0.f;
        const int r = tid / QR_LD, k = tid - r * QR_LD;                     // QR_ROWS * QR_LD = 272 <= QR_THREADS
        const bool in_x = r < rows && k < O, in_h = H > 0 && r < rows && k < H;
        if (in_x) xv = ((G1)p->obs)[(size_t)(r0 + r) * O + k];
        if (in_h) { hv = ((G1)p->h)[(size_t)(r0 + r) * H + k]; if (p->reset) rv = ((G1)p->reset)[r0 + r]; }
        float av = 1.f;                                                      // (no mask: every action available)
        const int t2 = tid - 512;                                            // waves 8..11: the mask, QR_ROWS * 64 entries
        const bool in_av = p->action && p->avail && t2 >= 0 && (t2 >> 6) < rows && (t2 & 63) < A;
        if (in_av) av = ((G1)p->avail)[(size_t)(r0 + (t2 >> 6)) * A + (t2 & 63)];
        if (tid < QR_ROWS * QR_LD) {
            lds[QR_X + tid] = xv;
            lds[QR_H + tid] = rv != 0.f ? 0.f : hv;
            lds[QR_A0 + tid] = 0.f; lds[QR_A1 + tid] = 0.f; lds[QR_HN + tid] = 0.f;
        }
        if (t2 >= 0 && t2 < QR_ROWS * 64) lds[QR_AV + t2] = av;
    }
    float eps = 0.f;
    if (p->action && tid < rows) eps = p->eps_dev ? *p->eps_dev : p->eps;
    // the step's coin and the row's uniform (marl_select_row's Philox draws) in two idle lanes, under the weight loads' latency
    if (p->action && tid >= QR_THREADS - 2) {
        const uint32_t step = p->step + (p->step_dev ? *p->step_dev : 0u);
        lds[QR_RNG + (tid & 1)] = (tid & 1) ? marl_step_coin(p->seed, step) : marl_row_uniform(p->seed, step, r0);
    }
    __syncthreads();
    QSTAMP(2);
    float v[RPW];
    // ---- W_hh h (needs nothing of the layers below)
    if (H > 0 && wave < 6 && l == n_pre + 1) {
        qr_product<RPW>(lds, QR_H, w, bias, v);
#pragma unroll
        for (int j = 0; j < RPW; ++j) if (j < rows) lds[QR_GH + j * QR_LDG + n] = v[j];
    }
    // ---- layers below the recurrence
    int in = QR_X;
    for (int i = 0; i < n_pre; ++i) {
        const int out = (i & 1) ? QR_A1 : QR_A0;
        if (wave == 6 + i && l == i) {
            qr_product<RPW>(lds, in, w, bias, v);
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                if (j < rows) lds[out + j * QR_LD + n] = ANYACT ? qf_act(v[j], act) : ((act == XRL_ACT_RELU && !(v[j] > 0.f)) ? 0.f : v[j]);
        }
        __syncthreads();
        in = out;
    }
    QSTAMP(3);
    if (H > 0) {
        if (wave < 6 && l == n_pre) {
            qr_product<RPW>(lds, in, w, bias, v);
#pragma unroll
            for (int j = 0; j < RPW; ++j) if (j < rows) lds[QR_GI + j * QR_LDG + n] = v[j];
        }
        __syncthreads();
        QSTAMP(4);
        if (tid < rows * H) {                                               // the cell (marl_act_gru_kernel's statements); rows * H <= 256
            const int r = tid / H, j = tid - r * H;
            const float* gi = lds + QR_GI + r * QR_LDG;
            const float* gh = lds + QR_GH + r * QR_LDG;
            const float h = lds[QR_H + r * QR_LD + j];
            const float hn = qa_cell(gi[j], gi[H + j], gi[2 * H + j], gh[j], gh[H + j], gh[2 * H + j], h);
            lds[QR_HN + r * QR_LD + j] = hn;
            p->h[(size_t)(r0 + r) * H + j] = hn;
        }
        __syncthreads();
        QSTAMP(5);
        in = QR_HN;
    }
    // ---- Q head
    for (int i = 0; i < n_post; ++i) {
        const bool last = i == n_post - 1;
        const int out = last ? QR_Q : (((n_pre + i) & 1) ? QR_A1 : QR_A0);
        if (wave == 9 + i && l >= 0) {
            qr_product<RPW>(lds, in, w, bias, v);
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                if (j < rows) {
                    const float y = last ? v[j] : (ANYACT ? qf_act(v[j], act) : ((act == XRL_ACT_RELU && !(v[j] > 0.f)) ? 0.f : v[j]));
                    lds[out + j * QR_LD + n] = y;
                    if (last) p->q[(size_t)(r0 + j) * p->ldq + n] = y;
                }
        }
        __syncthreads();
        in = out;
    }
    QSTAMP(6);
    if (p->action && tid < rows) {
        const int r = r0 + tid;
        const int a = marl_pick_row(lds + QR_Q + tid * QR_LD, lds + QR_AV + tid * 64, A, lds[QR_RNG + 1], lds[QR_RNG], eps);
        p->action[r] = a;
        if (p->action_f) p->action_f[r] = (float)a;
    }
    QSTAMP(7); QSTAMP(15);
}
#undef QSTAMP

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_qmix_fused_lds_bytes(const xrl_qmix_fused_t* p) {
    if (!p || p->n_layers < 1 || p->n_layers > XRL_QF_MAX_LAYERS || p->items_per_wg < 1) return -1;
    return qf_layout(*p).total * 4;
}

extern "C" int xrl_qmix_fused_layout(const xrl_qmix_fused_t* p, xrl_qf_image_t* out) {
    XRL_CHECK_ARG(p && out && p->n_layers >= 1 && p->n_layers <= XRL_QF_MAX_LAYERS);
    qf_image_layout(*p, *out);
    XRL_CHECK_ARG((out->agent_floats & 3) == 0 && (out->mixer_floats & 3) == 0);
    return XRL_OK;
}

extern "C" int xrl_qmix_fused_update(const xrl_qmix_fused_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_qmix_fused_t& p = *pp;
    XRL_CHECK_ARG(p.img_eval && p.img_target && ((reinterpret_cast<uintptr_t>(p.img_eval) | reinterpret_cast<uintptr_t>(p.img_target)) & 15) == 0);
    XRL_CHECK_ARG(p.obs && p.obs_next && p.state && p.state_next && p.actions && p.rewards &&
                  p.terminals && p.agent_mask && p.slabs && p.partials);
    XRL_CHECK_ARG(p.n_layers >= 1 && p.n_layers <= XRL_QF_MAX_LAYERS && p.B > 0 && p.items_per_wg > 0);
    XRL_CHECK_ARG(p.N >= 1 && p.N <= 64 && p.H >= 1 && p.H <= 64 && p.A >= 1 && p.dims[p.n_layers] == p.A && p.HH >= 1 && p.S >= 1);
    XRL_CHECK_ARG((p.slab_stride & 3) == 0);
    XRL_CHECK_ARG(p.ring_n_envs == 0 || (p.ring_n_envs > 0 && p.ring_n_size > 0 && p.size_dev != nullptr));
    const QfLds L = qf_layout(p);
    const size_t bytes = (size_t)L.total * 4;
    XRL_CHECK_ARG(bytes <= 160 * 1024);
    const bool mm = qf_mfma_products(p);
    static size_t allowed[2] = {0, 0};
    if (bytes > allowed[mm]) {
        if (mm) XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qmix_fused_kernel<true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        else {
            XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qmix_fused_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qmix_fused_kernel<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        }
        allowed[mm] = bytes;
    }
    const int n_wg = (p.B + p.items_per_wg - 1) / p.items_per_wg;
    QfArgs args{};
    xrl_qf_image_t im;
    qf_image_layout(p, im);
    args.p = p; args.L = L; args.agent4 = im.agent_floats / 4; args.mixer4 = im.mixer_floats / 4;
    const bool any_act = p.act != XRL_ACT_NONE && p.act != XRL_ACT_RELU;
    if (mm) hipLaunchKernelGGL((qmix_fused_kernel<true, true, false>), dim3(n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    else if (any_act) hipLaunchKernelGGL((qmix_fused_kernel<false, true, false>), dim3(n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    else hipLaunchKernelGGL((qmix_fused_kernel<false, false, false>), dim3(n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

// workgroups of the phase launch that one XCD keeps resident (one 1 024-thread workgroup per CU)
static int qf_phase_capacity() { const int c = device_cu_count() / 8; return c < QF_PHASE_MAX_WG ? c : QF_PHASE_MAX_WG; }

extern "C" int xrl_qmix_fused_phase_fits(int32_t B, int32_t items_per_wg, int64_t P) {
    if (B <= 0 || items_per_wg <= 0 || P <= 0 || (P & 3)) return 0;
    const int n_wg = (B + items_per_wg - 1) / items_per_wg;
    return n_wg <= qf_phase_capacity() ? 1 : 0;
}

extern "C" int xrl_qmix_fused_phase(const xrl_qmix_fused_t* pp, const xrl_qmix_phase_t* php, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr && php != nullptr);
    const xrl_qmix_fused_t& p = *pp;
    const xrl_qmix_phase_t& ph = *php;
    XRL_CHECK_ARG(p.img_eval && p.img_target && ((reinterpret_cast<uintptr_t>(p.img_eval) | reinterpret_cast<uintptr_t>(p.img_target)) & 15) == 0);
    XRL_CHECK_ARG(p.obs && p.obs_next && p.state && p.state_next && p.actions && p.rewards && p.terminals && p.agent_mask && p.slabs);
    XRL_CHECK_ARG(p.n_layers >= 1 && p.n_layers <= XRL_QF_MAX_LAYERS && p.B > 0 && p.items_per_wg > 0);
    XRL_CHECK_ARG(p.N >= 1 && p.N <= 64 && p.H >= 1 && p.H <= 64 && p.A >= 1 && p.dims[p.n_layers] == p.A && p.HH >= 1 && p.S >= 1);
    XRL_CHECK_ARG((p.slab_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(p.slabs) & 15) == 0);
    XRL_CHECK_ARG(p.ring_n_envs > 0 && p.ring_n_size > 0 && p.size_dev != nullptr);          // every update draws its own batch
    XRL_CHECK_ARG(ph.n_updates >= 1 && ph.params && ph.grad && ph.m && ph.v && ph.state && ph.map && ph.target && ph.phase_partials &&
                  ph.epoch_sums && ph.sumsq_part && ph.sync && ph.P > 0 && (ph.P & 3) == 0 && ph.P <= p.slab_stride);
    XRL_CHECK_ARG((ph.act_image == nullptr) == (ph.act_map == nullptr));
    XRL_CHECK_ARG(!qf_mfma_products(p));                                                      // (the VALU product instances)
    const int n_wg = (p.B + p.items_per_wg - 1) / p.items_per_wg;
    if (!xrl_qmix_fused_phase_fits(p.B, p.items_per_wg, ph.P)) {
        set_error("xrl_qmix_fused_phase: %d workgroups do not stay resident on one XCD of this device (%d)", n_wg, qf_phase_capacity());
        return XRL_EINVAL;
    }
    const QfLds L = qf_layout(p);
    const size_t bytes = (size_t)L.total * 4;
    XRL_CHECK_ARG(bytes <= 160 * 1024);
    xrl_qf_image_t im;
    qf_image_layout(p, im);
    const int per = (int)((ph.P / 4 + n_wg - 1) / n_wg);
    XRL_CHECK_ARG(32 * per + 2 * 20 <= im.agent_floats);                                          // the slab sums' LDS scratch (where the eval agent's weights were)
    static size_t allowed = 0;
    if (bytes > allowed) {
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qmix_fused_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qmix_fused_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        allowed = bytes;
    }
    XRL_CHECK_ARG(ph.scalars != nullptr && ph.n_updates <= 64);
    hipLaunchKernelGGL(qf_adam_scalars_kernel, dim3(1), dim3(64), 0, as_stream(stream), ph.state, ph.n_updates, ph.scalars);
    QfArgsPh args{};
    args.a.p = p; args.a.L = L; args.a.agent4 = im.agent_floats / 4; args.a.mixer4 = im.mixer_floats / 4;
    args.ph = ph;
    const bool any_act = p.act != XRL_ACT_NONE && p.act != XRL_ACT_RELU;
    if (any_act) hipLaunchKernelGGL((qmix_fused_kernel<false, true, true>), dim3(8 * n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    else hipLaunchKernelGGL((qmix_fused_kernel<false, false, true>), dim3(8 * n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_debug_act_gru_stamps(long long* stamps) {
    XRL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_qa_dbg), &stamps, sizeof(stamps)));
    return XRL_OK;
}

extern "C" int xrl_marl_act_gru_layout(const xrl_marl_act_gru_t* p, xrl_qa_image_t* out) {
    XRL_CHECK_ARG(p && out && p->n_pre >= 0 && p->n_post >= 1 && p->n_pre + p->n_post + 2 <= XRL_QA_MAX_LAYERS && p->rows_per_wg >= 1);
    const QaLds L = qa_layout(*p);
    for (int l = 0; l < XRL_QA_MAX_LAYERS; ++l) { out->w[l] = L.w[l]; out->b[l] = L.b[l]; out->ldw[l] = L.ldw[l]; }
    out->image_floats = L.image_floats;
    out->interleaved = qa_rows_form(*p) ? 1 : 0;
    out->lds_bytes = out->interleaved ? QR_TOTAL * 4 : L.total * 4;
    return XRL_OK;
}

extern "C" int xrl_marl_act_gru(const xrl_marl_act_gru_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_marl_act_gru_t& p = *pp;
    XRL_CHECK_ARG(p.image && p.obs && p.q && p.R > 0 && p.rows_per_wg > 0 && p.H >= 0 && p.O >= 1 && (p.h || p.H == 0));
    XRL_CHECK_ARG(p.n_pre >= 0 && p.n_post >= 1 && p.n_pre + p.n_post + 2 <= XRL_QA_MAX_LAYERS && (p.H > 0 || p.n_pre >= 1));
    XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(p.image) & 15) == 0 && p.ldq >= p.post[p.n_post - 1]);
    XRL_CHECK_ARG(p.action == nullptr || p.rows_per_wg <= QF_THREADS);
    QaArgs args{};
    args.p = p;
    args.L = qa_layout(p);
    XRL_CHECK_ARG((args.L.image_floats & 3) == 0);
    const int n_wg = (p.R + p.rows_per_wg - 1) / p.rows_per_wg;
    if (qa_rows_form(p)) {                                                 // one thread per output, weights in registers
        if (p.act != XRL_ACT_NONE && p.act != XRL_ACT_RELU) hipLaunchKernelGGL(marl_act_rows_kernel<true>, dim3(p.R), dim3(QR_THREADS), 0, as_stream(stream), args);
        else hipLaunchKernelGGL(marl_act_rows_kernel<false>, dim3(p.R), dim3(QR_THREADS), 0, as_stream(stream), args);
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    const size_t bytes = (size_t)args.L.total * 4;
    XRL_CHECK_ARG(bytes <= 160 * 1024);
    static size_t allowed = 0;
    if (bytes > allowed) {
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(marl_act_gru_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(marl_act_gru_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        allowed = bytes;
    }
    if (p.act != XRL_ACT_NONE && p.act != XRL_ACT_RELU) hipLaunchKernelGGL(marl_act_gru_kernel<true>, dim3(n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    else hipLaunchKernelGGL(marl_act_gru_kernel<false>, dim3(n_wg), dim3(QF_THREADS), bytes, as_stream(stream), args);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
