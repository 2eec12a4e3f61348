// Convolutions of the Atari networks (rl_models/representations/cnn.py:11-50 Basic_CNN, :53-102 AC_CNN_Atari;
// modules/layers.py:36-65 cnn_block: Conv2d(k, s, pad=(k-s)//2) + ReLU) as IMPLICIT GEMMs on the fp32 matrix cores:
// forward, gradient w.r.t. the layer input and weight gradient read the NHWC activations (the first layer: the uint8
// frames as the replay ring stores them) in place -- no column matrix is written or read back (csrc/conv.hip's
// im2col + GEMM pair stays as the any-shape path and as this file's test twin).
//
// Mapping.  The products are small (0.2-0.7 GFLOP per layer at batch 32: 300-1 300 strips of 32 output rows) and the chip has
// 1 024 SIMDs, so the unit of work is ONE WAVE = one 32-row strip x all N <= 64 output columns (N/32 accumulators of
// v_mfma_f32_32x32x2_f32), with no LDS staging and no barrier in the reduction loop:
//   * A operand: lane (row i, half h) loads the 4 consecutive channels (tap, c .. c+3) of ITS row's patch straight from the
//     image -- 16 bytes (float32) or one 32-bit word (the 4 stacked uint8 frames of one pixel) -- and feeds them to four
//     MFMAs; the reduction index is permuted to (th, tw, c) for that, which a sum allows as long as B uses the same order;
//   * B operand: a fragment-ordered image of the weights (xrl_gather_images): the 64 lanes of a wave read 1 KB contiguous;
//   * three register stages in flight per wave (loads of step q+2 issued before the MFMAs of step q);
//   * a layer with few strips (3 200 rows = 100 strips x 3 passes) is spread further by splitting the reduction over the 2 / 4
//     waves of a workgroup (partial accumulators meet in LDS once, fixed order).
// uint8 frames are scaled with a 256-entry LDS table of t / 255.0f (the reference divides, cnn.py:45; a table keeps the
// quotient bit-identical without a division per pixel).
#include "common.h"

namespace xrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CONV_MAX_GROUPS = 8;

struct ConvBatch {
    xrl_conv_t g[CONV_MAX_GROUPS];
    int n_groups, ks, n_split, pad;
    int64_t slab_stride;
};

template <int NB>
struct ConvStage {
    float4 a;                 // uint8 image: .x carries the raw 32-bit pixel
    float4 b[NB];
};

template <int NB, bool U8>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvBatch p) {
    __shared__ float s_tab[256];
    __shared__ __attribute__((aligned(16))) float s_red[3 * NB * 16 * 64];
    const xrl_conv_t& g = p.g[blockIdx.y];
    const int ks = p.ks, rs = 4 / ks;
    const int M = g.B * g.nh * g.nw, strips = (M + 31) >> 5;
    if ((int)blockIdx.x * rs >= strips) return;                          // (uniform: the grid is sized for the largest group)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x * rs + wave / ks, kpart = wave % ks;
    if (U8) s_tab[threadIdx.x] = (float)threadIdx.x / 255.0f;

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
    const int Q = (g.Th * Tw * C) >> 3;
    const int qper = (Q + ks - 1) / ks;
    const int q0 = kpart * qper, q1 = strip < strips ? min(Q, q0 + qper) : q0;
    const int twinv = (65536 + Tw - 1) / Tw;
    const char* __restrict__ img = reinterpret_cast<const char*>(g.img);
    const float4* __restrict__ wimg = reinterpret_cast<const float4*>(g.w);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    auto issue = [&](int q, ConvStage<NB>& st) {
        const int kq = 8 * q + 4 * lh;
        const int tap = kq >> lc, c = kq & (C - 1);
        const int th = (tap * twinv) >> 16, tw = tap - th * Tw;
        const int ih = bh + th, iw = bw + tw;
        const bool ok = rv && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
        const int pix = (b * IH + ih) * IW + iw;
        if (U8) {
            uint32_t v = 0;
            if (ok) v = *reinterpret_cast<const uint32_t*>(img + (size_t)pix * 4);
            st.a.x = __uint_as_float(v);
        } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(img + ((size_t)pix * C + c) * 4);
            st.a = v;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) st.b[nb] = wimg[(size_t)(q * NB + nb) * 64 + lane];
    };
    auto consume = [&](const ConvStage<NB>& st) {
        float a4[4];
        if (U8) {
            const uint32_t v = __float_as_uint(st.a.x);
#pragma unroll
            for (int s = 0; s < 4; ++s) a4[s] = s_tab[(v >> (8 * s)) & 255u];
        } else {
            a4[0] = st.a.x; a4[1] = st.a.y; a4[2] = st.a.z; a4[3] = st.a.w;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float b4[4] = {st.b[nb].x, st.b[nb].y, st.b[nb].z, st.b[nb].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc[nb], 0, 0, 0);
        }
    };

    ConvStage<NB> st0, st1, st2;
    int q = q0;
    if (q < q1) issue(q, st0);
    if (q + 1 < q1) issue(q + 1, st1);
    lds_barrier();                                                       // the table (global loads stay in flight)
    while (q < q1) {
        if (q + 2 < q1) issue(q + 2, st2);
        consume(st0);
        if (++q >= q1) break;
        if (q + 2 < q1) issue(q + 2, st0);
        consume(st1);
        if (++q >= q1) break;
        if (q + 2 < q1) issue(q + 2, st1);
        consume(st2);
        ++q;
    }

    if (ks > 1) {                                                        // the k-parts of a strip meet in LDS, part 0 adds in order
        const int slot = (wave / ks) * (ks - 1) + kpart - 1;
        if (kpart > 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s_red[((slot * NB + nb) * 16 + i) * 64 + lane] = acc[nb][i];
        }
        lds_barrier();
        if (kpart > 0) return;
        for (int kp = 1; kp < ks; ++kp) {
            const int sl = (wave / ks) * (ks - 1) + kp - 1;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nb][i] += s_red[((sl * NB + nb) * 16 + i) * 64 + lane];
        }
    }
    if (strip >= strips) return;

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); lane i < 32 knows
    // where row i of the strip goes
    const int my_off = rv ? ((b * g.OHt + hh * g.so + g.ph) * g.OWt + ww * g.so + g.pw) * g.N : -1;
    const float* __restrict__ mask = g.mask;
    const bool relu = g.act == XRL_ACT_RELU;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = nb * 32 + li;
        const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
            const int off = __shfl(my_off, row, 64);
            if (off < 0) continue;
            float v = acc[nb][i] + bias;
            if (relu) v = v > 0.f ? v : 0.f;
            if (mask) v = mask[off + col] > 0.f ? v : 0.f;
            g.out[off + col] = v;
        }
    }
}

// Weight gradient.  Workgroup = (group, 32 reduction-index columns k' = (th, tw, c), row chunk `split`); its four waves take a
// quarter of the chunk's rows each and meet in LDS.  MFMA A operand = dY^T (lane (f, h): dY[row_h][f], 128 contiguous bytes
// per row across the lanes), B operand = the patch value (row_h, k') read from the image in place.
template <int NB, bool U8>
__global__ void __launch_bounds__(256) conv_dw_mfma_kernel(ConvBatch p) {
    __shared__ float s_tab[256];
    __shared__ __attribute__((aligned(16))) float s_red[4 * NB * 16 * 64];
    __shared__ float s_db[4 * NB * 64];
    const xrl_conv_t& g = p.g[blockIdx.z];
    const int C = g.C, lc = 31 - __clz(C), Tw = g.Tw, kk = g.Th * Tw, Kp = kk * C;
    const int ktile = blockIdx.x, split = blockIdx.y;
    if (ktile * 32 >= Kp) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    if (U8) s_tab[threadIdx.x] = (float)threadIdx.x / 255.0f;
    const int M = g.B * g.nh * g.nw;
    const int chunk = ((M + p.n_split - 1) / p.n_split + 1) & ~1;
    const int r_beg = split * chunk, r_end = min(M, r_beg + chunk);
    const int sub = ((chunk + 3) / 4 + 1) & ~1;
    const int w_beg = r_beg + wave * sub, w_end = min(r_end, w_beg + sub);

    const int kq = ktile * 32 + li;
    const int tap = kq >> lc, c = kq & (C - 1);
    const int th = tap / Tw, tw = tap - th * Tw;
    const int N = g.N, IH = g.IH, IW = g.IW, nh = g.nh, nw = g.nw, sh = g.sh;
    const int oh_t = g.off_h + th, ow_t = g.off_w + tw;

    int row = w_beg + lh, b, hh, ww;
    {
        const int per = nh * nw;
        b = row / per;
        const int rem = row - b * per;
        hh = rem / nw;
        ww = rem - hh * nw;
    }
    const char* __restrict__ img = reinterpret_cast<const char*>(g.img);
    const float* __restrict__ dy = g.dy;

    f32x16 acc[NB];
    float dbs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        dbs[nb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    }
    lds_barrier();
    constexpr int U = 4;
    for (int base = w_beg; base < w_end; base += 2 * U) {
        float av[U][NB], bv[U];
        uint32_t braw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool okr = row < w_end;
            const int ih = hh * sh + oh_t, iw = ww * sh + ow_t;
            const bool ok = okr && (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
            const int pix = (b * IH + ih) * IW + iw;
            if (U8) {
                braw[u] = 0;
                if (ok) braw[u] = *reinterpret_cast<const uint8_t*>(img + (size_t)pix * 4 + c);
            } else {
                bv[u] = 0.f;
                if (ok) bv[u] = *reinterpret_cast<const float*>(img + ((size_t)pix * C + c) * 4);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) av[u][nb] = okr ? dy[(size_t)row * N + nb * 32 + li] : 0.f;
            row += 2; ww += 2;
            if (ww >= nw) { ww -= nw; if (++hh >= nh) { hh = 0; ++b; } }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float bf = U8 ? s_tab[braw[u]] : bv[u];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][nb], bf, acc[nb], 0, 0, 0);
                dbs[nb] += av[u][nb];
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s_red[((wave * NB + nb) * 16 + i) * 64 + lane] = acc[nb][i];
        s_db[(wave * NB + nb) * 64 + lane] = dbs[nb];
    }
    __syncthreads();
    // wave w adds registers 4w .. 4w+3 of the four partial tiles (fixed order) and stores them in the reference layout
    float* __restrict__ dw = g.out + (size_t)split * p.slab_stride;
    const int col_off = c * kk + tap;                                    // [N][C][Th][Tw]: k' = (tap, c) -> c * kk + tap
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * wave + j;
            float s = s_red[((0 * NB + nb) * 16 + i) * 64 + lane];
            s += s_red[((1 * NB + nb) * 16 + i) * 64 + lane];
            s += s_red[((2 * NB + nb) * 16 + i) * 64 + lane];
            s += s_red[((3 * NB + nb) * 16 + i) * 64 + lane];
            const int f = nb * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            dw[(size_t)f * Kp + col_off] = s;
        }
    if (ktile == 0 && g.dbias && threadIdx.x < NB * 32) {
        const int nb = threadIdx.x >> 5, f = threadIdx.x & 31;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += s_db[(w * NB + nb) * 64 + f] + s_db[(w * NB + nb) * 64 + 32 + f];
        g.dbias[(size_t)split * p.slab_stride + nb * 32 + f] = s;
    }
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
    XRL_CHECK_ARG(pow2(g.C) && g.C >= 4 && (!g.img_u8 || g.C == 4));
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

extern "C" int xrl_conv_fwd(const xrl_conv_t* groups, int n_groups, int k_split, xrl_stream_t stream) {
    XRL_CHECK_ARG(groups && n_groups >= 1 && n_groups <= CONV_MAX_GROUPS && (k_split == 1 || k_split == 2 || k_split == 4));
    ConvBatch b;
    b.n_groups = n_groups; b.ks = k_split; b.n_split = 1; b.pad = 0; b.slab_stride = 0;
    int max_strips = 0;
    for (int i = 0; i < n_groups; ++i) {
        const int rc = check_group(groups[i], false);
        if (rc != XRL_OK) return rc;
        XRL_CHECK_ARG(groups[i].N == groups[0].N && !groups[i].img_u8 == !groups[0].img_u8);
        b.g[i] = groups[i];
        const int strips = (int)(((int64_t)groups[i].B * groups[i].nh * groups[i].nw + 31) / 32);
        max_strips = strips > max_strips ? strips : max_strips;
    }
    const int rs = 4 / k_split;
    const dim3 grid((max_strips + rs - 1) / rs, n_groups);
    const bool u8 = groups[0].img_u8 != 0;
    if (groups[0].N == 32) {
        if (u8) hipLaunchKernelGGL((conv_mfma_kernel<1, true>), grid, dim3(256), 0, as_stream(stream), b);
        else hipLaunchKernelGGL((conv_mfma_kernel<1, false>), grid, dim3(256), 0, as_stream(stream), b);
    } else {
        if (u8) hipLaunchKernelGGL((conv_mfma_kernel<2, true>), grid, dim3(256), 0, as_stream(stream), b);
        else hipLaunchKernelGGL((conv_mfma_kernel<2, false>), grid, dim3(256), 0, as_stream(stream), b);
    }
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_conv_bwd_weight(const xrl_conv_t* groups, int n_groups, int n_split, int64_t slab_stride, xrl_stream_t stream) {
    XRL_CHECK_ARG(groups && n_groups >= 1 && n_groups <= CONV_MAX_GROUPS && n_split >= 1 && n_split <= 65535);
    // the kernel is instantiated per (N / 32, uint8): groups that differ go out as separate launches (same stream, any order)
    for (int nbv = 1; nbv <= 2; ++nbv)
        for (int u8 = 0; u8 <= 1; ++u8) {
            ConvBatch b;
            b.n_groups = 0; b.ks = 1; b.n_split = n_split; b.pad = 0; b.slab_stride = slab_stride;
            int max_tiles = 0;
            for (int i = 0; i < n_groups; ++i) {
                const int rc = check_group(groups[i], true);
                if (rc != XRL_OK) return rc;
                if (groups[i].N != 32 * nbv || (groups[i].img_u8 != 0) != (u8 != 0)) continue;
                b.g[b.n_groups++] = groups[i];
                const int tiles = groups[i].Th * groups[i].Tw * groups[i].C / 32;
                max_tiles = tiles > max_tiles ? tiles : max_tiles;
            }
            if (!b.n_groups) continue;
            const dim3 grid(max_tiles, n_split, b.n_groups);
            if (nbv == 1) {
                if (u8) hipLaunchKernelGGL((conv_dw_mfma_kernel<1, true>), grid, dim3(256), 0, as_stream(stream), b);
                else hipLaunchKernelGGL((conv_dw_mfma_kernel<1, false>), grid, dim3(256), 0, as_stream(stream), b);
            } else {
                if (u8) hipLaunchKernelGGL((conv_dw_mfma_kernel<2, true>), grid, dim3(256), 0, as_stream(stream), b);
                else hipLaunchKernelGGL((conv_dw_mfma_kernel<2, false>), grid, dim3(256), 0, as_stream(stream), b);
            }
            XRL_CHECK_LAUNCH();
        }
    return XRL_OK;
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
