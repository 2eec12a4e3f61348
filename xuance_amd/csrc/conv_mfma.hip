// Convolutions of the Atari networks (rl_models/representations/cnn.py:11-50 Basic_CNN, :53-102 AC_CNN_Atari;
// modules/layers.py:36-65 cnn_block: Conv2d(k, s, pad=(k-s)//2) + ReLU) as IMPLICIT GEMMs on the fp32 matrix cores:
// forward, gradient w.r.t. the layer input and weight gradient read the NHWC activations (the first layer: the uint8
// frames as the replay ring stores them) in place -- no column matrix is written or read back (csrc/conv.hip's
// im2col + GEMM pair stays as the any-shape path and as this file's test twin).
//
// Mapping.  The products are small (0.2-0.7 GFLOP per layer at batch 32: 300-1 300 strips of 32 output rows) and the chip has
// 1 024 SIMDs, so the unit of work is ONE WAVE = one 32-row strip x all N <= 64 output columns (N/32 accumulators of
// v_mfma_f32_32x32x2_f32), with no LDS staging and no barrier in the reduction loop:
//   * A operand: lane (row i, half h) loads 16 consecutive reduction indices of ITS row's patch straight from the image -- 16
//     channels of one tap (float32: 64 bytes) or 4 pixels x the 4 stacked uint8 frames (16 bytes) -- and feeds them to 16
//     MFMAs; the reduction index is permuted to (th, tw, c) for that, which a sum allows as long as B uses the same order;
//   * B operand: a fragment-ordered image of the weights (xrl_gather_images): the 64 lanes of a wave read 1 KB contiguous;
//   * 3-4 register stages (groups of 32 reduction indices) in flight per wave;
//   * a layer with few strips (3 200 rows = 100 strips x 3 passes) is spread further by splitting the reduction over 2 / 4 / 8
//     waves of a workgroup; the partial accumulators meet in LDS once and every wave finishes a share of the tile.
// uint8 frames are scaled with a 256-entry LDS table of t / 255.0f (the reference divides, cnn.py:45; a table keeps the
// quotient bit-identical without a division per pixel).
#include "common.h"

namespace xrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));

// Operands are read with buffer loads: 32-bit byte offsets from a scalar base (no 64-bit address arithmetic in the vector ALU --
// a first version spent ~150 cycles of integer multiplies per load instruction), and an offset past the end returns 0, which
// is exactly what a tap outside the image or a row past the end has to contribute.
constexpr unsigned CONV_OOB = 0x7fffff00u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

constexpr int CONV_MAX_GROUPS = 8;

struct ConvBatch {
    xrl_conv_t g[CONV_MAX_GROUPS];
    int n_groups, ks, n_split, pad;
    int64_t slab_stride;
    long long* dbg;           // tools/probe_conv_phases.py: per workgroup [real-time start, end, shader clock at 6 phase marks]
};

template <int NB>
struct ConvStage {            // operands of ONE group of 32 reduction indices (four MFMA sub-steps of 8)
    float4 a[4];              // float32 image: the lane's 16 contiguous channels; uint8 image: a[0] = four raw 32-bit pixels
    float4 b[4][NB];
};

// (the wave index goes through readfirstlane everywhere in this file: what is derived from it -- the wave's reduction range, hence
// the scalar offset of its weight loads -- is then uniform for the compiler too; without it every such load sat in a waterfall loop)
// KS waves share one 32-row strip and split its reduction; a workgroup has 4 (KS <= 4) or 8 waves = 4 / KS or 1 strips.
template <int NB, bool U8, int KS, int D>
__global__ void __launch_bounds__(KS == 8 ? 512 : 256) conv_mfma_kernel(ConvBatch p) {
    constexpr int WAVES = KS == 8 ? 8 : 4, RS = WAVES / KS;
    __shared__ float s_tab[256];
    extern __shared__ __attribute__((aligned(16))) float s_red[];       // KS > 1: [WAVES][NB][16][64]
    kernarg_prefetch<(sizeof(ConvBatch) < 1024 ? sizeof(ConvBatch) : 1024)>();
    const xrl_conv_t& g = p.g[blockIdx.y];
    const int M = g.B * g.nh * g.nw, strips = (M + 31) >> 5;
    if ((int)blockIdx.x * RS >= strips) return;                          // (uniform: the grid is sized for the largest group)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x * RS + wave / KS, kpart = wave % KS;
    long long* dbg = p.dbg ? p.dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
    const bool dbg_me = dbg && threadIdx.x == 0;
#define CONV_STAMP(k) do { if (dbg_me) dbg[k] = clock64(); } while (0)
    if (dbg_me) dbg[0] = (long long)__builtin_amdgcn_s_memrealtime();
    CONV_STAMP(2);
    if (U8 && threadIdx.x < 256) s_tab[threadIdx.x] = (float)threadIdx.x / 255.0f;

    const int r = strip * 32 + li;
    const bool rv = strip < strips && r < M;
    int b = 0, hh = 0, ww = 0;
    if (rv) {
        const int per = g.nh * g.nw;
        b = r / per;
        const int rem = r - b * per;
        hh = rem / g.nw;
        ww = rem - hh * g.nw;
    }
    const int bh = hh * g.sh + g.off_h, bw = ww * g.sh + g.off_w;
    const int C = g.C, lc = 31 - __clz(C), Tw = g.Tw, IH = g.IH, IW = g.IW;
    const int G = (g.Th * Tw * C) >> 5;
    const int gper = (G + KS - 1) / KS;
    const int g0 = kpart * gper, g1 = strip < strips ? min(G, g0 + gper) : g0;
    const int twinv = (65536 + Tw - 1) / Tw;
    const int pbase = (b * IH + bh) * IW + bw;                           // pixel index of tap (0, 0) (may lie outside: checked per tap)
    const __amdgpu_buffer_rsrc_t irs = conv_rsrc(g.img, (unsigned)(g.B * IH * IW * C) * (U8 ? 1u : 4u));
    const __amdgpu_buffer_rsrc_t wrs = conv_rsrc(g.w, (unsigned)(g.N * g.Th * Tw * C) * 4u);
    float bias[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias[nb] = g.bias ? g.bias[nb * 32 + li] : 0.f;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    // A lane takes 16 CONTIGUOUS reduction indices of its row per group -- k' = 32 gq + 16 (lane / 32) + 4 s + j for sub-step s,
    // MFMA j -- so that the two lanes of a row read one whole 128-byte line of the image between them in four back-to-back
    // loads (a first version gave a lane 4 indices per step: every load instruction touched 64 different lines for 16 bytes
    // each, the 16 KB L1 kept none of them until the next step, and the loop ran at one L2 request per lane and step).
    // Loads are UNCONDITIONAL (clamped group; invalid taps read from an out-of-range offset = 0): with a load under a branch
    // the compiler cannot count the loads in flight at the point of use and waits for all of them (s_waitcnt vmcnt(0) in
    // front of every step).
    auto issue = [&](int gq, ConvStage<NB>& st) {
        const bool live = gq < g1;                                       // (uniform) past the wave's range: every offset out of range,
        gq = min(gq, G - 1);                                             // the load instructions stay, nothing is fetched
        const int kq = 32 * gq + 16 * lh;
        const int tap = kq >> lc, c = kq & (C - 1);
        const int th = __mul24(tap, twinv) >> 16, tw = tap - __mul24(th, Tw);
        const int ih = bh + th;
        const bool rowok = live && rv && (unsigned)ih < (unsigned)IH;
        const int pix = pbase + __mul24(th, IW) + tw;
        if (U8) {                                                        // C == 4: the lane's 16 indices = 4 pixels of one kernel row
            unsigned raw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = rowok && (unsigned)(bw + tw + j) < (unsigned)IW;
                raw[j] = __builtin_amdgcn_raw_buffer_load_b32(irs, ok ? (unsigned)(pix + j) << 2 : CONV_OOB, 0, 0);
            }
            st.a[0] = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3]));
        } else {                                                         // C >= 16: 16 channels of one tap
            const bool ok = rowok && (unsigned)(bw + tw) < (unsigned)IW;
            const unsigned off = ok ? ((unsigned)pix << (lc + 2)) + ((unsigned)c << 2) : CONV_OOB;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const cu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, off + 16 * s, 0, 0);
                st.a[s] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const cu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, live ? lane * 16 : CONV_OOB, ((gq * 4 + s) * NB + nb) * 1024, 0);
                st.b[s][nb] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
    };
    auto consume = [&](const ConvStage<NB>& st) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a4[4];
            if (U8) {
                const unsigned v = __float_as_uint(s == 0 ? st.a[0].x : (s == 1 ? st.a[0].y : (s == 2 ? st.a[0].z : st.a[0].w)));
#pragma unroll
                for (int j = 0; j < 4; ++j) a4[j] = s_tab[(v >> (8 * j)) & 255u];
            } else {
                a4[0] = st.a[s].x; a4[1] = st.a[s].y; a4[2] = st.a[s].z; a4[3] = st.a[s].w;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float b4[4] = {st.b[s][nb].x, st.b[s][nb].y, st.b[s][nb].z, st.b[s][nb].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc[nb], 0, 0, 0);
            }
        }
    };

    // D register stages: the loads of group gq + D - 1 go out before the MFMAs of group gq
    ConvStage<NB> st[D];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(g0 + d, st[d]);
    CONV_STAMP(3);
    lds_barrier();                                                       // the table (global loads stay in flight)
    for (int gq = g0; gq < g1; gq += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue(gq + d + D - 1, st[(d + D - 1) % D]);
            if (gq + d < g1) consume(st[d]);
        }
    }
    CONV_STAMP(4);

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); lane i < 32 knows
    // where row i of the strip goes.  KS > 1: the partial tiles of a strip meet in LDS and each of its KS waves finishes
    // (fixed-order sum, bias, activation, mask, store) 16 NB / KS of the 16 NB accumulator registers.
    const int my_off = rv ? ((b * g.OHt + hh * g.so + g.ph) * g.OWt + ww * g.so + g.pw) * g.N : -1;
    const float* __restrict__ mask = g.mask;
    const bool relu = g.act == XRL_ACT_RELU;
    constexpr int T = 16 * NB / KS;
    int offs[T];
    float mk[T], val[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int pr = kpart + KS * t, i = pr & 15;
        offs[t] = __shfl(my_off, (i & 3) + 8 * (i >> 2) + 4 * lh, 64);
    }
    if (mask) {                                                          // (uniform; every mask value requested before the first use)
#pragma unroll
        for (int t = 0; t < T; ++t) mk[t] = mask[(offs[t] < 0 ? 0 : offs[t]) + ((kpart + KS * t) >> 4) * 32 + li];
    }
    if (KS > 1) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s_red[((wave * NB + nb) * 16 + i) * 64 + lane] = acc[nb][i];
        lds_barrier();
        CONV_STAMP(5);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int pr = kpart + KS * t;                               // = nb * 16 + i
            const int w0 = wave - kpart;
            float v = s_red[((w0 * NB) * 16 + pr) * 64 + lane];
#pragma unroll
            for (int kp = 1; kp < KS; ++kp) v += s_red[(((w0 + kp) * NB) * 16 + pr) * 64 + lane];
            val[t] = v;
        }
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) val[t] = acc[t >> 4][t & 15];
    }
    if (strip >= strips) return;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int nb = (kpart + KS * t) >> 4;
        float v = val[t] + (NB == 1 ? bias[0] : (nb ? bias[NB - 1] : bias[0]));
        if (relu) v = v > 0.f ? v : 0.f;
        if (mask) v = mk[t] > 0.f ? v : 0.f;
        if (offs[t] >= 0) g.out[offs[t] + nb * 32 + li] = v;
    }
    if (dbg_me) {
        dbg[6] = clock64();
        dbg[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#undef CONV_STAMP
}

// Weight gradient.  Workgroup = (group, 32 reduction-index columns k' = (th, tw, c), row chunk `split`); its eight waves take an
// eighth of the chunk's rows each -- 25..55 rows, ALL of whose operands are requested before the first product: the loop is a
// chain of memory round trips otherwise (measured: 7 batches x ~4 k cycles per wave) -- and meet in LDS.  MFMA A operand = dY^T (lane (f, h): dY[row_h][f], 128 contiguous bytes
// per row across the lanes), B operand = the patch value (row_h, k') read from the image in place.
// the kernel's body: a routine so that one launch can carry groups of different (N / 32, uint8) kinds (conv_dw_mixed_kernel below)
template <int NB, bool U8>
__device__ __forceinline__ void conv_dw_body(const ConvBatch& p, float* s_tab, float* s_red, float* s_db) {
    constexpr int DW_WAVES = 8;
    const xrl_conv_t& g = p.g[blockIdx.z];
    const int C = g.C, lc = 31 - __clz(C), Tw = g.Tw, kk = g.Th * Tw, Kp = kk * C;
    const int ktile = blockIdx.x, split = blockIdx.y;
    if (ktile * 32 >= Kp) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    if (U8 && threadIdx.x < 256) s_tab[threadIdx.x] = (float)threadIdx.x / 255.0f;
    long long* dbg = p.dbg ? p.dbg + 8 * ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) : nullptr;
    const bool dbg_me = dbg && threadIdx.x == 0;
    if (dbg_me) { dbg[0] = (long long)__builtin_amdgcn_s_memrealtime(); dbg[2] = clock64(); }
    const int M = g.B * g.nh * g.nw;
    const int n_split = g.pad > 0 ? g.pad : p.n_split;                   // (xrl_conv_t.pad: this group's own number of row chunks)
    if (split >= n_split) return;
    const int chunk = ((M + n_split - 1) / n_split + 1) & ~1;
    const int r_beg = split * chunk, r_end = min(M, r_beg + chunk);
    const int sub = ((chunk + DW_WAVES - 1) / DW_WAVES + 1) & ~1;
    const int w_beg = r_beg + wave * sub, w_end = min(r_end, w_beg + sub);

    const int kq = ktile * 32 + li;
    const int tap = kq >> lc, c = kq & (C - 1);
    const int th = tap / Tw, tw = tap - th * Tw;
    const int N = g.N, lN = 31 - __clz(N), IH = g.IH, IW = g.IW, nh = g.nh, nw = g.nw, sh = g.sh;
    const int oh_t = g.off_h + th, ow_t = g.off_w + tw;

    int row = w_beg + lh, b, hh, ww;
    {
        const int per = nh * nw;
        b = row / per;
        const int rem = row - b * per;
        hh = rem / nw;
        ww = rem - hh * nw;
    }
    // pixel index of this lane's tap for (b, hh, ww = 0), kept current with additions only
    int rowbase = (b * IH + hh * sh + oh_t) * IW + ow_t;
    const int d_row = sh * IW, d_frame = (IH - nh * sh) * IW;
    const __amdgpu_buffer_rsrc_t irs = conv_rsrc(g.img, (unsigned)(g.B * IH * IW * C) * (U8 ? 1u : 4u));
    const __amdgpu_buffer_rsrc_t drs = conv_rsrc(g.dy, (unsigned)(M * N) * 4u);

    f32x16 acc[NB];
    float dbs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        dbs[nb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    }
    lds_barrier();
    // batches of U row pairs, two register sets: the loads of batch i + 1 are in flight while batch i is multiplied; rows past
    // the end and taps outside the image read 0 from an out-of-range offset (unconditional loads, see above)
    constexpr int U = 16;
    struct Batch { float av[U][NB]; float bv[U]; unsigned braw[U]; };
    auto load = [&](Batch& t) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool okr = row < w_end;
            const int wo = __mul24(ww, sh);
            const bool ok = okr && (unsigned)(__mul24(hh, sh) + oh_t) < (unsigned)IH && (unsigned)(wo + ow_t) < (unsigned)IW;
            const int pix = rowbase + wo;
            if (U8) t.braw[u] = __builtin_amdgcn_raw_buffer_load_b8(irs, ok ? ((unsigned)pix << 2) + c : CONV_OOB, 0, 0);
            else t.bv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(irs, ok ? ((unsigned)pix << (lc + 2)) + ((unsigned)c << 2) : CONV_OOB, 0, 0));
            const unsigned doff = okr ? ((unsigned)row << (lN + 2)) + ((unsigned)li << 2) : CONV_OOB;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) t.av[u][nb] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(drs, doff + 128 * nb, 0, 0));
            row += 2; ww += 2;
            if (ww >= nw) {
                ww -= nw; rowbase += d_row;
                if (++hh >= nh) { hh = 0; ++b; rowbase += d_frame; }
            }
        }
    };
    auto multiply = [&](const Batch& t) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float bf = U8 ? s_tab[t.braw[u] & 255u] : t.bv[u];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.av[u][nb], bf, acc[nb], 0, 0, 0);
                dbs[nb] += t.av[u][nb];
            }
        }
    };
    Batch t0, t1;
    load(t0);
    if (dbg_me) dbg[3] = clock64();
    for (int base = w_beg; base < w_end; base += 4 * U) {
        load(t1);
        multiply(t0);
        load(t0);
        if (base + 2 * U < w_end) multiply(t1);
    }
    if (dbg_me) dbg[4] = clock64();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s_red[((wave * NB + nb) * 16 + i) * 64 + lane] = acc[nb][i];
        s_db[(wave * NB + nb) * 64 + lane] = dbs[nb];
    }
    __syncthreads();
    // wave w adds registers 2w, 2w+1 of the eight partial tiles (fixed order) and stores them in the reference layout
    float* __restrict__ dw = g.out + (size_t)split * p.slab_stride;
    const int col_off = c * kk + tap;                                    // [N][C][Th][Tw]: k' = (tap, c) -> c * kk + tap
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 16 / DW_WAVES; ++j) {
            const int i = (16 / DW_WAVES) * wave + j;
            float s = s_red[((0 * NB + nb) * 16 + i) * 64 + lane];
#pragma unroll
            for (int w = 1; w < DW_WAVES; ++w) s += s_red[((w * NB + nb) * 16 + i) * 64 + lane];
            const int f = nb * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            dw[(size_t)f * Kp + col_off] = s;
        }
    if (ktile == 0 && g.dbias && threadIdx.x < NB * 32) {
        const int nb = threadIdx.x >> 5, f = threadIdx.x & 31;
        float s = 0.f;
        for (int w = 0; w < DW_WAVES; ++w) s += s_db[(w * NB + nb) * 64 + f] + s_db[(w * NB + nb) * 64 + 32 + f];
        g.dbias[(size_t)split * p.slab_stride + nb * 32 + f] = s;
    }
    if (dbg_me) { dbg[5] = clock64(); dbg[1] = (long long)__builtin_amdgcn_s_memrealtime(); }
}

template <int NB, bool U8>
__global__ void __launch_bounds__(512) conv_dw_mfma_kernel(ConvBatch p) {
    __shared__ float s_tab[256];
    extern __shared__ __attribute__((aligned(16))) float s_red[];        // [DW_WAVES][NB][16][64]
    __shared__ float s_db[8 * NB * 64];
    conv_dw_body<NB, U8>(p, s_tab, s_red, s_db);
}

// Groups of different kinds in ONE launch (round 6): the weight gradients of a convolution stack are independent of each other, but the
// first layer (uint8 frames, 32 filters) and the upper ones (float32 activations, 64 filters) are different instances of the body and
// went out as two launches one after the other (DQN-C3: 10.5 + 16.8 us, each on a part of the chip).  The kind is uniform per workgroup
// (blockIdx.z = group).
__global__ void __launch_bounds__(512) conv_dw_mixed_kernel(ConvBatch p) {
    __shared__ float s_tab[256];
    extern __shared__ __attribute__((aligned(16))) float s_red[];        // sized for NB = 2
    __shared__ float s_db[8 * 2 * 64];
    const xrl_conv_t& g = p.g[blockIdx.z];
    const bool two = g.N == 64, u8 = g.img_u8 != 0;
    if (two) { if (u8) conv_dw_body<2, true>(p, s_tab, s_red, s_db); else conv_dw_body<2, false>(p, s_tab, s_red, s_db); }
    else { if (u8) conv_dw_body<1, true>(p, s_tab, s_red, s_db); else conv_dw_body<1, false>(p, s_tab, s_red, s_db); }
}

struct ImageJobs {
    xrl_image_job_t j[8];
    int n;
};

__global__ void __launch_bounds__(256) gather_images_kernel(ImageJobs p) {
    const xrl_image_job_t& j = p.j[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.n; i += (int64_t)gridDim.x * 256) {
        const int m = j.map[i];
        j.dst[i] = m >= 0 ? j.src[m] : 0.f;
    }
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int check_group(const xrl_conv_t& g, bool weight_grad) {
    XRL_CHECK_ARG(g.img && g.out && g.B > 0 && g.IH > 0 && g.IW > 0 && g.Th > 0 && g.Tw > 0 && g.Tw <= 16 && g.nh > 0 && g.nw >= 2);
    XRL_CHECK_ARG(pow2(g.C) && (g.img_u8 ? g.C == 4 && g.Tw % 4 == 0 : g.C >= (weight_grad ? 4 : 16)));
    XRL_CHECK_ARG(g.N == 32 || g.N == 64);
    XRL_CHECK_ARG((g.Th * g.Tw * g.C) % 32 == 0 && g.Th * g.Tw < 4096);
    XRL_CHECK_ARG((int64_t)g.B * g.IH * g.IW * g.C < ((int64_t)1 << 31));
    if (weight_grad) {
        XRL_CHECK_ARG(g.dy != nullptr);
        XRL_CHECK_ARG((int64_t)g.B * g.nh * g.nw * g.N < ((int64_t)1 << 31));
    } else {
        XRL_CHECK_ARG(g.w != nullptr && g.sh > 0 && g.so > 0 && g.OHt > 0 && g.OWt > 0);
        XRL_CHECK_ARG((int64_t)g.B * g.OHt * g.OWt * g.N < ((int64_t)1 << 31));
        XRL_CHECK_ARG((reinterpret_cast<uintptr_t>(g.w) & 15) == 0 && (g.img_u8 || (reinterpret_cast<uintptr_t>(g.img) & 15) == 0));
    }
    return XRL_OK;
}

}  // namespace xrl

using namespace xrl;

static int conv_fwd_launch(const xrl_conv_t* groups, int n_groups, int k_split, long long* dbg, xrl_stream_t stream) {
    XRL_CHECK_ARG(groups && n_groups >= 1 && n_groups <= CONV_MAX_GROUPS && (k_split == 1 || k_split == 2 || k_split == 4 || k_split == 8));
    ConvBatch b;
    b.n_groups = n_groups; b.ks = k_split; b.n_split = 1; b.pad = 0; b.slab_stride = 0; b.dbg = dbg;
    int max_strips = 0;
    for (int i = 0; i < n_groups; ++i) {
        const int rc = check_group(groups[i], false);
        if (rc != XRL_OK) return rc;
        XRL_CHECK_ARG(groups[i].N == groups[0].N && !groups[i].img_u8 == !groups[0].img_u8);
        b.g[i] = groups[i];
        const int strips = (int)(((int64_t)groups[i].B * groups[i].nh * groups[i].nw + 31) / 32);
        max_strips = strips > max_strips ? strips : max_strips;
    }
    const int threads = k_split == 8 ? 512 : 256, rs = (threads / 64) / k_split;
    const dim3 grid((max_strips + rs - 1) / rs, n_groups);
    const bool u8 = groups[0].img_u8 != 0;
    const size_t lds = k_split > 1 ? (size_t)(threads / 64) * (groups[0].N / 32) * 16 * 64 * sizeof(float) : 0;
    static bool attr_done = false;
#define CONV_INST(NBV, U8V, KSV, DV)                                                                                         \
    {                                                                                                                        \
        if (!attr_done)                                                                                                      \
            XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<NBV, U8V, KSV, DV>),            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NBV * 16 * 64 * 4));           \
        if (groups[0].N == 32 * NBV && u8 == U8V && k_split == KSV)                                                          \
            hipLaunchKernelGGL((conv_mfma_kernel<NBV, U8V, KSV, DV>), grid, dim3(threads), lds, as_stream(stream), b);       \
    }
#define CONV_INST_KS(NBV, U8V, DV) CONV_INST(NBV, U8V, 1, DV) CONV_INST(NBV, U8V, 2, DV) CONV_INST(NBV, U8V, 4, DV) CONV_INST(NBV, U8V, 8, DV)
    CONV_INST_KS(1, true, 4)
    CONV_INST_KS(1, false, 3)
    CONV_INST_KS(2, true, 3)
    CONV_INST_KS(2, false, 3)
#undef CONV_INST_KS
#undef CONV_INST
    attr_done = true;
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_conv_fwd(const xrl_conv_t* groups, int n_groups, int k_split, xrl_stream_t stream) {
    return conv_fwd_launch(groups, n_groups, k_split, nullptr, stream);
}
// diagnostics (tools/probe_conv_phases.py): the same launch with per-workgroup time stamps, 8 x int64 per workgroup
extern "C" int xrl_conv_fwd_probe(const xrl_conv_t* groups, int n_groups, int k_split, long long* dbg, xrl_stream_t stream) {
    return conv_fwd_launch(groups, n_groups, k_split, dbg, stream);
}

// OFF by default: measured slower (profiles/r06_g_conv_overlap.json: DQN-C3 update 120.7 -> 125.8 us, PPO-Atari update 7.61 -> 8.11 ms) -- the
// merged grid is sized for the largest group and the 212-register body halves the residency of the 32-filter groups
static bool g_conv_dw_mixed = false;         // xrl_set_conv_dw_mixed (tests, A/B measurements)
extern "C" int xrl_set_conv_dw_mixed(int enable) { g_conv_dw_mixed = enable != 0; return XRL_OK; }

static int conv_dw_launch(const xrl_conv_t* groups, int n_groups, int n_split, int64_t slab_stride, long long* dbg, xrl_stream_t stream) {
    XRL_CHECK_ARG(groups && n_groups >= 1 && n_groups <= CONV_MAX_GROUPS && n_split >= 1 && n_split <= 65535);
    // the body is instantiated per (N / 32, uint8).  Groups of more than one kind: ONE launch of the mixed kernel (g_conv_dw_mixed, round 6);
    // else (or with the switch off) one launch per kind (same stream, any order)
    {
        int kinds = 0, max_tiles = 0;
        ConvBatch b;
        b.n_groups = 0; b.ks = 1; b.n_split = n_split; b.pad = 0; b.slab_stride = slab_stride; b.dbg = dbg;
        for (int i = 0; i < n_groups; ++i) {
            const int rc = check_group(groups[i], true);
            if (rc != XRL_OK) return rc;
            XRL_CHECK_ARG(groups[i].N == 32 || groups[i].N == 64);
            XRL_CHECK_ARG(groups[i].pad >= 0 && groups[i].pad <= n_split);
            kinds |= 1 << ((groups[i].N == 64 ? 2 : 0) + (groups[i].img_u8 ? 1 : 0));
            b.g[b.n_groups++] = groups[i];
            const int tiles = groups[i].Th * groups[i].Tw * groups[i].C / 32;
            max_tiles = tiles > max_tiles ? tiles : max_tiles;
        }
        if (g_conv_dw_mixed && (kinds & (kinds - 1)) != 0 && dbg == nullptr) {
            static bool attr_mixed = false;
            if (!attr_mixed) {
                XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 16 * 64 * 4));
                attr_mixed = true;
            }
            hipLaunchKernelGGL(conv_dw_mixed_kernel, dim3(max_tiles, n_split, b.n_groups), dim3(512), (size_t)8 * 2 * 16 * 64 * sizeof(float), as_stream(stream), b);
            XRL_CHECK_LAUNCH();
            return XRL_OK;
        }
    }
    for (int nbv = 1; nbv <= 2; ++nbv)
        for (int u8 = 0; u8 <= 1; ++u8) {
            ConvBatch b;
            b.n_groups = 0; b.ks = 1; b.n_split = n_split; b.pad = 0; b.slab_stride = slab_stride; b.dbg = dbg;
            int max_tiles = 0;
            for (int i = 0; i < n_groups; ++i) {
                const int rc = check_group(groups[i], true);
                if (rc != XRL_OK) return rc;
                if (groups[i].N != 32 * nbv || (groups[i].img_u8 != 0) != (u8 != 0)) continue;
                XRL_CHECK_ARG(groups[i].pad >= 0 && groups[i].pad <= n_split);
                b.g[b.n_groups++] = groups[i];
                const int tiles = groups[i].Th * groups[i].Tw * groups[i].C / 32;
                max_tiles = tiles > max_tiles ? tiles : max_tiles;
            }
            if (!b.n_groups) continue;
            const dim3 grid(max_tiles, n_split, b.n_groups);
            const size_t lds = (size_t)8 * nbv * 16 * 64 * sizeof(float);
            static bool attr_done = false;
            if (!attr_done) {
                XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_mfma_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 16 * 64 * 4));
                XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_mfma_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 16 * 64 * 4));
                attr_done = true;
            }
            if (nbv == 1) {
                if (u8) hipLaunchKernelGGL((conv_dw_mfma_kernel<1, true>), grid, dim3(512), lds, as_stream(stream), b);
                else hipLaunchKernelGGL((conv_dw_mfma_kernel<1, false>), grid, dim3(512), lds, as_stream(stream), b);
            } else {
                if (u8) hipLaunchKernelGGL((conv_dw_mfma_kernel<2, true>), grid, dim3(512), lds, as_stream(stream), b);
                else hipLaunchKernelGGL((conv_dw_mfma_kernel<2, false>), grid, dim3(512), lds, as_stream(stream), b);
            }
            XRL_CHECK_LAUNCH();
        }
    return XRL_OK;
}

extern "C" int xrl_conv_bwd_weight(const xrl_conv_t* groups, int n_groups, int n_split, int64_t slab_stride, xrl_stream_t stream) {
    return conv_dw_launch(groups, n_groups, n_split, slab_stride, nullptr, stream);
}
extern "C" int xrl_conv_bwd_weight_probe(const xrl_conv_t* groups, int n_groups, int n_split, int64_t slab_stride, long long* dbg,
                                         xrl_stream_t stream) {
    return conv_dw_launch(groups, n_groups, n_split, slab_stride, dbg, stream);
}

extern "C" int xrl_gather_images(const xrl_image_job_t* jobs, int n_jobs, xrl_stream_t stream) {
    XRL_CHECK_ARG(jobs && n_jobs >= 1 && n_jobs <= 8);
    ImageJobs p;
    p.n = n_jobs;
    int64_t nmax = 0;
    for (int i = 0; i < n_jobs; ++i) {
        XRL_CHECK_ARG(jobs[i].src && jobs[i].map && jobs[i].dst && jobs[i].n > 0);
        p.j[i] = jobs[i];
        nmax = jobs[i].n > nmax ? jobs[i].n : nmax;
    }
    int64_t nb = (nmax + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(gather_images_kernel, dim3((unsigned)nb, n_jobs), dim3(256), 0, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
