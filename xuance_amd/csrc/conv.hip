// Convolution stack of the Atari Q-network (rl_models/representations/cnn.py:11-50, modules/layers.py:36-65:
// x/255, Conv2d(k, s, pad=(k-s)//2) + ReLU, AdaptiveMaxPool2d((1,1))) as im2col + the engine's fp32-MFMA GEMMs.
// Activations stay NHWC end to end: the GEMM output [B*OH*OW][F] of one layer IS the NHWC input of the next, and the
// im2col column order (c, kh, kw) makes the reference's weight tensor [F][C][kh][kw] the GEMM's B operand as stored.
//   forward :  col = im2col(x);  y = relu(col . W^T + b)                       (xrl_linear_fwd)
//   backward:  dW = dY^T . col, db = colsum(dY)  (xrl_linear_bwd_weight);  dcol = dY . W  (xrl_linear_bwd_data);
//              dX = col2im(dcol) * [x > 0]  (gather form: every input element sums its patches in a fixed order)
// These kernels are HBM-bound copies: im2col writes K/(s*s*C) times the input once, col2im reads it back.
#include "common.h"

namespace xrl {

// One workgroup per group of output positions (b, oh, ow): the position is decomposed once per row (scalar), each thread
// then handles the column entries j = (c, kh, kw) of that row with 32-bit arithmetic and coalesced stores.  (The first
// version decomposed a flat 64-bit element index per thread -- five 64-bit divisions per element -- and ran at 0.75 TB/s.)
template <typename T>
__global__ void __launch_bounds__(256) im2col_nhwc_kernel(const T* __restrict__ x, float* __restrict__ col, int B, int H, int W,
                                                          int C, int k, int s, int p, int OH, int OW, int scale255) {
    const int K = C * k * k, kk = k * k;
    const int rows = B * OH * OW;
    constexpr int RPB = 4;                                               // rows per workgroup
    if (sizeof(T) == 1 && C == 4 && kk == 64) {
        // Atari first layer (uint8, 4 stacked frames, 8x8 kernel): one wave per output position, lane = (kh, kw) loads
        // the 4 channels of its pixel with ONE 32-bit load (each patch row is 32 contiguous bytes) and writes them to the
        // four channel planes of the column row: every store instruction of the wave covers 256 contiguous bytes.
        const int row = blockIdx.x * RPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= rows) return;
        const int ow = row % OW, t = row / OW, oh = t % OH, b = t / OH;
        const int kh = lane >> 3, kw = lane & 7;
        const int ih = oh * s - p + kh, iw = ow * s - p + kw;
        uint32_t px = 0;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W)
            px = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(x) + (((size_t)b * H + ih) * W + iw) * 4);
        float* out = col + (size_t)row * K + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = (float)((px >> (8 * c)) & 0xffu);
            if (scale255) v = v / 255.0f;
            out[c * 64] = v;
        }
        return;
    }
    for (int r = 0; r < RPB; ++r) {
        const int row = blockIdx.x * RPB + r;
        if (row >= rows) return;
        const int ow = row % OW, t = row / OW, oh = t % OH, b = t / OH;
        const int ih0 = oh * s - p, iw0 = ow * s - p;
        const T* xb = x + (size_t)b * H * W * C;
        float* out = col + (size_t)row * K;
        for (int j = threadIdx.x; j < K; j += 256) {
            const int c = j / kk, rem = j - c * kk, kh = rem / k, kw = rem - kh * k;
            const int ih = ih0 + kh, iw = iw0 + kw;
            float v = 0.f;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                v = (float)xb[((size_t)ih * W + iw) * C + c];
                if (scale255) v = v / 255.0f;                            // observations / 255.0 (cnn.py:45)
            }
            out[j] = v;
        }
    }
}

// dx[b][h][w][c] = sum_{kh,kw} dcol[(b, oh, ow)][c*k*k + kh*k + kw]  over the patches that cover (h, w); times [xact > 0]
__global__ void __launch_bounds__(256) col2im_nhwc_kernel(const float* __restrict__ dcol, const float* __restrict__ xact,
                                                          float* __restrict__ dx, int B, int H, int W, int C, int k, int s, int p,
                                                          int OH, int OW) {
    const int K = C * k * k;
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float acc = 0.f;
        if (!xact || xact[i] > 0.f) {
            for (int kh = 0; kh < k; ++kh) {
                const int th = h + p - kh;
                if (th < 0 || th % s) continue;
                const int oh = th / s;
                if (oh >= OH) continue;
                for (int kw = 0; kw < k; ++kw) {
                    const int tw = w + p - kw;
                    if (tw < 0 || tw % s) continue;
                    const int ow = tw / s;
                    if (ow >= OW) continue;
                    acc += dcol[(((int64_t)b * OH + oh) * OW + ow) * K + c * k * k + kh * k + kw];
                }
            }
        }
        dx[i] = acc;
    }
}

// AdaptiveMaxPool2d((1,1)) over the P = OH*OW positions of y[B][P][F]; first maximum wins (argmax kept for backward).
// One workgroup per frame: thread (part, f) scans every (256/F)-th position, the parts meet in LDS (larger value, then
// smaller position, so the result is the first maximum whatever the split).
__global__ void __launch_bounds__(256) maxpool_hw_fwd_kernel(const float* __restrict__ y, float* __restrict__ feat,
                                                             int32_t* __restrict__ arg, int B, int P, int F, int ld_feat) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int b = blockIdx.x;
    const int parts = 256 / F > 0 ? 256 / F : 1;
    for (int f0 = 0; f0 < F; f0 += 256) {                               // F > 256: several passes
        const int f = f0 + (threadIdx.x % (F < 256 ? F : 256)), part = threadIdx.x / (F < 256 ? F : 256);
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (f < F && part < parts) {
            const float* src = y + (int64_t)b * P * F + f;
            constexpr int U = 8;                                        // 8 loads in flight per thread (the scan is latency-bound)
            int q = part;
            for (; q + (U - 1) * parts < P; q += U * parts) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = src[(int64_t)(q + u * parts) * F];
#pragma unroll
                for (int u = 0; u < U; ++u) if (v[u] > best) { best = v[u]; bi = q + u * parts; }
            }
            for (; q < P; q += parts) { const float v = src[(int64_t)q * F]; if (v > best) { best = v; bi = q; } }
        }
        s_v[threadIdx.x] = best; s_i[threadIdx.x] = bi;
        __syncthreads();
        if (part == 0 && f < F) {
            const int w = F < 256 ? F : 256;
            for (int q = 1; q < parts; ++q) {
                const float v = s_v[q * w + (f - f0)];
                const int vi = s_i[q * w + (f - f0)];
                if (v > best || (v == best && vi < bi)) { best = v; bi = vi; }
            }
            feat[(int64_t)b * ld_feat + f] = best;
            if (arg) arg[b * F + f] = bi;
        }
        __syncthreads();
    }
}

// dY[b][q][f] = dfeat[b][f] if q == argmax[b][f] and y > 0 (ReLU) else 0
__global__ void __launch_bounds__(256) maxpool_hw_bwd_kernel(const float* __restrict__ dfeat, const int32_t* __restrict__ arg,
                                                             const float* __restrict__ y, float* __restrict__ dy, int B, int P,
                                                             int F, int ld_dfeat) {
    const int64_t total = (int64_t)B * P * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int q = (int)((i / F) % P), b = (int)(i / ((int64_t)F * P));
        float v = 0.f;
        if (arg[b * F + f] == q && y[i] > 0.f) v = dfeat[(int64_t)b * ld_dfeat + f];
        dy[i] = v;
    }
}

// nn.Flatten() of an NCHW activation (AC_CNN_Atari, cnn.py:90): the engine keeps conv outputs NHWC -- y[(b, q), f] with q = h * W
// + w -- while the dense layer behind the flatten indexes its input (f, q) = f * P + q.  One workgroup per frame, the frame's
// P x F block goes through LDS so that both the read and the write are contiguous.
// (round 6: 1 024 threads per frame and the loads of a pass requested four at a time before the first LDS store -- as plain loops
//  of 256 threads every element was a global round trip of its own, twelve in a row per thread: 12 / 18 us per launch at 256 frames)
constexpr int FLAT_THREADS = 1024;
__global__ void __launch_bounds__(FLAT_THREADS) flatten_chw_fwd_kernel(const float* __restrict__ y, float* __restrict__ feat, int P, int F,
                                                                       int ld_feat) {
    extern __shared__ float s_blk[];                                    // [P][F + 1]
    const int b = blockIdx.x, n = P * F;
    const float* src = y + (int64_t)b * n;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * FLAT_THREADS) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * FLAT_THREADS; v[u] = i < n ? src[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * FLAT_THREADS; if (i < n) s_blk[(i / F) * (F + 1) + i % F] = v[u]; }
    }
    __syncthreads();
    float* dst = feat + (int64_t)b * ld_feat;
    for (int i = threadIdx.x; i < n; i += FLAT_THREADS) dst[i] = s_blk[(i % P) * (F + 1) + i / P];    // i = f * P + q
}

// dY[(b, q), f] = dfeat[b][f * P + q] where y > 0 (the ReLU in front of the flatten), else 0
__global__ void __launch_bounds__(FLAT_THREADS) flatten_chw_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ y,
                                                                       float* __restrict__ dy, int P, int F, int ld_dfeat) {
    extern __shared__ float s_blk[];                                    // [F][P + 1]
    const int b = blockIdx.x, n = P * F;
    const float* src = dfeat + (int64_t)b * ld_dfeat;
    const float* yy = y + (int64_t)b * n;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * FLAT_THREADS) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * FLAT_THREADS; v[u] = i < n ? src[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * FLAT_THREADS; if (i < n) s_blk[(i / P) * (P + 1) + i % P] = v[u]; }    // i = f * P + q
    }
    __syncthreads();
    float* dst = dy + (int64_t)b * n;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * FLAT_THREADS) {
        float yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * FLAT_THREADS; yv[u] = i < n ? yy[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FLAT_THREADS;
            if (i < n) dst[i] = yv[u] > 0.f ? s_blk[(i % F) * (P + 1) + i / F] : 0.f;   // i = q * F + f
        }
    }
}

static int grid_for(int64_t total) {
    int64_t nb = (total + 255) / 256;
    return (int)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb));
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_im2col_nhwc(const void* x, int x_is_u8, float* col, int B, int H, int W, int C, int k, int s, int p,
                               xrl_stream_t stream) {
    XRL_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0 && p >= 0);
    const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    XRL_CHECK_ARG(OH > 0 && OW > 0);
    XRL_CHECK_ARG((int64_t)B * OH * OW < ((int64_t)1 << 31));
    const unsigned grid = (unsigned)(((int64_t)B * OH * OW + 3) / 4);                 // 4 output positions per workgroup
    if (x_is_u8)
        hipLaunchKernelGGL(im2col_nhwc_kernel<uint8_t>, dim3(grid), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const uint8_t*>(x), col, B, H, W, C, k, s, p, OH, OW, 1);
    else
        hipLaunchKernelGGL(im2col_nhwc_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const float*>(x), col, B, H, W, C, k, s, p, OH, OW, 0);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_col2im_nhwc(const float* dcol, const float* xact, float* dx, int B, int H, int W, int C, int k, int s, int p,
                               xrl_stream_t stream) {
    XRL_CHECK_ARG(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0 && p >= 0);
    const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    hipLaunchKernelGGL(col2im_nhwc_kernel, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, as_stream(stream), dcol, xact, dx,
                       B, H, W, C, k, s, p, OH, OW);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_maxpool_hw_fwd(const float* y, float* feat, int32_t* argmax, int B, int P, int F, int ld_feat,
                                  xrl_stream_t stream) {
    XRL_CHECK_ARG(y && feat && B > 0 && P > 0 && F > 0 && ld_feat >= F);
    hipLaunchKernelGGL(maxpool_hw_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), y, feat, argmax, B, P, F,
                       ld_feat);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_maxpool_hw_bwd(const float* dfeat, const int32_t* argmax, const float* y, float* dy, int B, int P, int F,
                                  int ld_dfeat, xrl_stream_t stream) {
    XRL_CHECK_ARG(dfeat && argmax && y && dy && B > 0 && P > 0 && F > 0 && ld_dfeat >= F);
    hipLaunchKernelGGL(maxpool_hw_bwd_kernel, dim3(grid_for((int64_t)B * P * F)), dim3(256), 0, as_stream(stream), dfeat, argmax, y,
                       dy, B, P, F, ld_dfeat);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_flatten_chw_fwd(const float* y, float* feat, int B, int P, int F, int ld_feat, xrl_stream_t stream) {
    XRL_CHECK_ARG(y && feat && B > 0 && P > 0 && F > 0 && ld_feat >= P * F && (size_t)P * (F + 1) * 4 <= 64 * 1024);
    hipLaunchKernelGGL(flatten_chw_fwd_kernel, dim3(B), dim3(FLAT_THREADS), (size_t)P * (F + 1) * 4, as_stream(stream), y, feat, P, F, ld_feat);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_flatten_chw_bwd(const float* dfeat, const float* y, float* dy, int B, int P, int F, int ld_dfeat, xrl_stream_t stream) {
    XRL_CHECK_ARG(dfeat && y && dy && B > 0 && P > 0 && F > 0 && ld_dfeat >= P * F && (size_t)F * (P + 1) * 4 <= 64 * 1024);
    hipLaunchKernelGGL(flatten_chw_bwd_kernel, dim3(B), dim3(FLAT_THREADS), (size_t)F * (P + 1) * 4, as_stream(stream), dfeat, y, dy, P, F, ld_dfeat);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
