// LDS-resident 32-row MLP tile on the fp32 matrix cores (shared by rollout_fused.hip and ppo_fused.hip).
// See rollout_fused.hip for the mapping notes.
#pragma once
#include "common.h"

namespace xrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FT = 32;               // rows per tile
constexpr int FUSED_THREADS = 512;   // 8 waves
constexpr int NW = FUSED_THREADS / 64;
constexpr int PD = 16;               // prefetched B chunks (of 8 k) held in registers for the first big layer
constexpr int SMALL_W = 4608;        // a layer whose padded weights fit in this many floats lives in the LDS cache

__host__ __device__ inline int level_ld(int width) { return ((width + 7) / 8) * 8 + 4; }
__host__ __device__ inline bool layer_small(int N, int K) { return N * level_ld(K) <= SMALL_W; }

#ifdef XRL_TILE_PROBE
static __device__ long long* g_probe = nullptr;
static __device__ int g_probe_i = 0;
#endif

// Slot of k-chunk q inside tile t of a fragment-ordered weight copy.  The waves of a workgroup fetch chunk q of their
// tiles at the same moment; consecutive tiles are a multiple of 4 KB apart, which is the period of the 16 x 256-byte L2
// channel interleave, so without the rotation every wave (of every CU) would hammer the same 4 channels.
__host__ __device__ inline int frag_slot(int q, int t, int n_chunks, int step) { return (q + t * step) % n_chunks; }

#define MFMA4(a, b, acc)                                                         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).x, (b).x, acc, 0, 0, 0);      \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).y, (b).y, acc, 0, 0, 0);      \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).z, (b).z, acc, 0, 0, 0);      \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a).w, (b).w, acc, 0, 0, 0);

// Narrow layer (N <= 8, e.g. the merged policy/value heads) on the VALU: thread (row = tid>>4, sub = tid&15) accumulates a
// 1/16 slice of K for every output column, the 16 partials are combined with xor-shuffles.  One barrier instead of the
// three of the split-K MFMA path; the summation order differs from the MFMA chain by fp32 rounding only.
__device__ __forceinline__ void narrow_layer_valu(const float* Wl, int ldw, const float* bias_l, int K, int N, int act,
                                                  const float* in, int ld_in, float* out, int ld_out) {
#ifdef XRL_TILE_PROBE
    if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64();
#endif
    const int r = threadIdx.x >> 4, sub = threadIdx.x & 15;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    const int k4 = (K + 3) / 4;                                         // float4 chunks; padding columns are zero on both sides
    for (int q = sub; q < k4; q += 16) {
        const float4 a = *reinterpret_cast<const float4*>(in + r * ld_in + q * 4);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < N) {
                const float4 w = *reinterpret_cast<const float4*>(Wl + c * ldw + q * 4);
                acc[c] += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c < N) {
            float v = acc[c];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if (sub == 0) { XRL_ACT_DISPATCH(act, out[r * ld_out + c] = act_apply_c<ACT>(v + bias_l[c]);) }
        }
    }
#ifdef XRL_TILE_PROBE
    if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64();
#endif
    __syncthreads();
#ifdef XRL_TILE_PROBE
    if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64();
#endif
}

// out[32][N] = act(in[32][K] . W[N][K]^T + bias)   in/out: LDS tiles.  Wl != null: weights in the LDS cache with row
// stride ldw (zero padded); else Wg in global memory, and pf[] holds this wave's first PD chunks when `use_pf`.
__device__ __forceinline__ void fused_layer(const float* __restrict__ Wg, const float* Wl, int ldw, const float* bias_l,
                                            int K, int N, int act, const float* in, int ld_in, float* out, int ld_out,
                                            float* red, const float4 (&pf)[PD], bool use_pf,
                                            const float* aux = nullptr, int ld_aux = 0, int tile_begin = 0, int tile_end = -1) {
    // tile_begin/tile_end restrict the output column tiles [tile_begin, tile_end) (a workgroup that owns only one branch
    // of a stacked actor|critic layer); with a restriction the waves map 1:1 onto the tiles (no split-K).
    // aux == null : out = act(acc + bias)                      (forward)
    // aux != null : out = acc * act'(aux[row][col])            (backward w.r.t. the layer input; `act` is the
    //               activation that produced aux, bias_l is unused)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
#ifdef XRL_TILE_PROBE
    if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64();
#endif
    const bool restricted = tile_end >= 0;
    const int n_tiles = restricted ? tile_end : (N + 31) / 32;
    const int kq = (K + 7) / 8;
    const bool fast_g = ((K & 7) == 0) && ((reinterpret_cast<uintptr_t>(Wg) & 15) == 0);
    const int wpt = (restricted || n_tiles >= NW) ? 1 : NW / n_tiles;   // waves per tile (split-K factor)
    const int tiles_per_pass = NW / wpt;
    for (int t0 = restricted ? tile_begin : 0; t0 < n_tiles; t0 += tiles_per_pass) {
        const int tile = t0 + wave / wpt, ks = wave % wpt;
        const bool live = tile < n_tiles && (wave / wpt) < tiles_per_pass;
        const int n0 = tile * 32;
        const int wr = min(n0 + li, N - 1);   // rows >= N only feed output columns that are never stored: clamp
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (live) {
            const float* arow = in + li * ld_in + 4 * lh;
            if (Wl) {
                const float* wrow = Wl + wr * ldw + 4 * lh;
#pragma unroll 4
                for (int q = ks; q < kq; q += wpt) {
                    const float4 b = *reinterpret_cast<const float4*>(wrow + q * 8);
                    const float4 a = *reinterpret_cast<const float4*>(arow + q * 8);
                    MFMA4(a, b, acc)
                }
            } else if (fast_g) {
                const float* wrow = Wg + (size_t)wr * K + 4 * lh;
                int qstart = 0;
                if (use_pf && t0 == (restricted ? tile_begin : 0) && wpt == 1) {   // chunks 0..PD-1 are already in registers
                    float4 af[PD];                                  // all A fragments first: one LDS latency, then pure MFMA
#pragma unroll
                    for (int q = 0; q < PD; ++q) af[q] = q < kq ? *reinterpret_cast<const float4*>(arow + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < PD; ++q) {
                        if (q < kq) { MFMA4(af[q], pf[q], acc) }
                    }
                    qstart = PD;
                }
#pragma unroll 8
                for (int q = qstart + ks; q < kq; q += wpt) {
                    const float4 b = *reinterpret_cast<const float4*>(wrow + q * 8);
                    const float4 a = *reinterpret_cast<const float4*>(arow + q * 8);
                    MFMA4(a, b, acc)
                }
            } else {
                const float* wrow = Wg + (size_t)wr * K;
                for (int q = ks; q < kq; q += wpt) {
                    const int kk = q * 8 + 4 * lh;
                    const float4 a = *reinterpret_cast<const float4*>(arow + q * 8);
                    float4 b;
                    b.x = kk + 0 < K ? wrow[kk + 0] : 0.f;
                    b.y = kk + 1 < K ? wrow[kk + 1] : 0.f;
                    b.z = kk + 2 < K ? wrow[kk + 2] : 0.f;
                    b.w = kk + 3 < K ? wrow[kk + 3] : 0.f;
                    MFMA4(a, b, acc)
                }
            }
        }
#ifdef XRL_TILE_PROBE
        { float t_ = acc[15]; asm volatile("" ::"v"(t_)); if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64(); }
#endif
        // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (wpt == 1) {
            const int col = n0 + li;
            if (live && col < N) {
                if (aux) {
                    XRL_ACT_DISPATCH(act,
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                            out[row * ld_out + col] = acc[r] * act_grad_c<ACT>(aux[row * ld_aux + col]);
                        })
                } else {
                    const float bv = bias_l[col];
                    XRL_ACT_DISPATCH(act,
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                            out[row * ld_out + col] = act_apply_c<ACT>(acc[r] + bv);
                        })
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                red[(wave * 32 + row) * 33 + li] = acc[r];
            }
            __syncthreads();
                XRL_ACT_DISPATCH(act,
                for (int i = threadIdx.x; i < tiles_per_pass * 32 * 32; i += FUSED_THREADS) {
                    const int tl = i >> 10, row = (i >> 5) & 31, c = i & 31;
                    const int col = (t0 + tl) * 32 + c;
                    if (t0 + tl < n_tiles && col < N) {
                        float v = 0.f;
                        for (int w = 0; w < wpt; ++w) v += red[((tl * wpt + w) * 32 + row) * 33 + c];   // fixed order
                        out[row * ld_out + col] = aux ? v * act_grad_c<ACT>(aux[row * ld_aux + col])
                                                      : act_apply_c<ACT>(v + bias_l[col]);
                    }
                })
            __syncthreads();
        }
    }
#ifdef XRL_TILE_PROBE
    if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64();
#endif
    __syncthreads();
#ifdef XRL_TILE_PROBE
    if (g_probe && threadIdx.x == 0 && (int)blockIdx.x == (int)g_probe[47]) g_probe[g_probe_i++] = clock64();
#endif
}


}  // namespace xrl
