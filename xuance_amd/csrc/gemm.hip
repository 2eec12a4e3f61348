// Dense layers on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
// Replaces nn.Linear forward / autograd backward of mlp_block (xuance/torch/rl_models/modules/layers.py:16-33)
// for the actor / critic / Q / hyper-network MLPs of the three learners.
//
// Tiling (64-wide wavefronts): one 256-thread workgroup = 4 waves in a 2x2 arrangement computes a 64x64 tile
// of C; each wave owns one 32x32 MFMA accumulator (16 VGPRs).  K is consumed in slabs of 32 staged through LDS.
// Three operand layouts are needed and they only differ in how the LDS slab is read:
//   NT  fwd         C[M,N]  = A[M,K]  . B[N,K]^T   both operands k-contiguous  -> ds_read_b128 of 4 k's per lane
//   NN  bwd_data    C[M,N]  = A[M,K]  . B[K,N]     B is k-major                -> ds_read_b32 rows
//   TN  bwd_weight  C[N',K'] = A[M,N']^T . X[M,K']  both k(=m)-major            -> ds_read_b32 rows, split over M
// The MFMA consumes, per instruction, k = {h} for lane-half h = lane>>5; since the GEMM sums over k any fixed
// permutation of k shared by A and B is legal, so the b128 path feeds lane-half h with k = 8q+4h+s (s = 0..3).
#include "common.h"

namespace xrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDK = BK + 4;   // k-contiguous slab row stride (floats): 144 B, conflict-free for ds_read_b128
constexpr int LDR = 64 + 0;   // k-major slab row stride (floats)
constexpr int MAX_GROUPS = 8;

struct GemmBatch {
    xrl_gemm_t g[MAX_GROUPS];
    int n_groups;
    int n_split;
    int64_t slab_stride;
};

enum { MODE_NT = 0, MODE_NN = 1, MODE_TN = 2 };

// ---- global -> register staging -------------------------------------------------------------------
// Operands are read with BUFFER loads whose offset is out of range for a row / column outside the matrix: the hardware
// returns 0 and fetches nothing, and -- the point -- every load is unconditional.  (The first version guarded its loads with
// branches (row < R, vector or scalar path, `if (next slab exists)`); the compiler cannot count the loads in flight across such
// branches and put s_waitcnt vmcnt(0) in front of every slab: the three-slot register ring below never had more than one slab
// in flight.  Found with the convolution kernels of csrc/conv_mfma.hip, same cause.)
typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned GEMM_OOB = 0x7fffff00u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_rsrc(const float* base, int rows, int ld, int cols) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)(((rows - 1) * ld + cols) * 4), 0x00020000);
}
__device__ __forceinline__ float4 as_f4(gu32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// k-contiguous operand: tile rows r0..r0+63 (bounded by R), k range k0..k0+31 (bounded by K); row-major, ld.
// thread t loads 2 float4: rows (t>>3) and (t>>3)+32, k offset (t&7)*4.  VEC: ld, K multiples of 4 and a 16-byte aligned base
// (one 16-byte load per quad; K % 4 == 0 makes every quad all-in or all-out); otherwise four 4-byte loads.
template <bool VEC>
__device__ __forceinline__ void load_kcontig(__amdgpu_buffer_rsrc_t rs, int ld, int R, int K, int r0, int k0, float4 (&reg)[2]) {
    const int t = threadIdx.x;
    const int kk = k0 + (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = r0 + (t >> 3) + 32 * i;
        const unsigned base = (__umul24(r, ld) + kk) << 2;
        if (VEC) {
            reg[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, (r < R && kk < K) ? base : GEMM_OOB, 0, 0));
        } else {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                e[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (r < R && kk + j < K) ? base + 4 * j : GEMM_OOB, 0, 0));
            reg[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}

// k-major operand: tile k rows k0..k0+31 (bounded by Kr), columns c0..c0+63 (bounded by C); row-major, ld.
// `ones_col`: column index that reads as 1.0 for valid rows (bias-gradient trick of the TN mode), or -1.
// thread t loads 2 float4: k rows (t>>4) and (t>>4)+16, column offset (t&15)*4.
template <bool VEC>
__device__ __forceinline__ void load_kmajor(__amdgpu_buffer_rsrc_t rs, int ld, int Kr, int C, int k0, int c0, int ones_col,
                                            float4 (&reg)[2]) {
    const int t = threadIdx.x;
    const int cc = c0 + (t & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = k0 + (t >> 4) + 16 * i;
        const unsigned base = (__umul24(k, ld) + cc) << 2;
        float4 v;
        if (VEC) {
            v = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rs, (k < Kr && cc < C) ? base : GEMM_OOB, 0, 0));
        } else {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                e[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (k < Kr && cc + j < C) ? base + 4 * j : GEMM_OOB, 0, 0));
            v = make_float4(e[0], e[1], e[2], e[3]);
        }
        if (ones_col >= 0) {
            const bool kv = k < Kr;
            if (kv && cc + 0 == ones_col) v.x = 1.f;
            if (kv && cc + 1 == ones_col) v.y = 1.f;
            if (kv && cc + 2 == ones_col) v.z = 1.f;
            if (kv && cc + 3 == ones_col) v.w = 1.f;
        }
        reg[i] = v;
    }
}

__device__ __forceinline__ void stage_kcontig(float* S, const float4 (&reg)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        *reinterpret_cast<float4*>(&S[((t >> 3) + 32 * i) * LDK + (t & 7) * 4]) = reg[i];
}
__device__ __forceinline__ void stage_kmajor(float* S, const float4 (&reg)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        *reinterpret_cast<float4*>(&S[((t >> 4) + 16 * i) * LDR + (t & 15) * 4]) = reg[i];
}

template <int MODE, bool VA, bool VB>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmBatch p) {
    __shared__ __attribute__((aligned(16))) float sA[64 * LDK > 32 * LDR ? 64 * LDK : 32 * LDR];
    __shared__ __attribute__((aligned(16))) float sB[64 * LDK > 32 * LDR ? 64 * LDK : 32 * LDR];

    kernarg_prefetch<sizeof(GemmBatch)>();
    const xrl_gemm_t& g = p.g[blockIdx.z];
    // logical problem: C[Mo, No] = sum_k opA[Mo,k] opB[k,No], k < Kred
    int Mo, No, Kred;
    if (MODE == MODE_TN) { Mo = g.N; No = g.K + (g.dbias ? 1 : 0); Kred = g.M; }
    else { Mo = g.M; No = g.N; Kred = g.K; }
    const int tiles_n = (No + BN - 1) / BN;
    const int tiles_m = (Mo + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_m * tiles_n) return;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // reduction range (split over the batch for the weight gradient; split-K of a forward layer, see xrl_linear_fwd)
    int kbeg = 0, kend = Kred;
    if (MODE == MODE_TN) {
        const int chunk = (Kred + p.n_split - 1) / p.n_split;
        kbeg = blockIdx.y * chunk;
        kend = min(Kred, kbeg + chunk);
    }
    const bool splitk = MODE == MODE_NT && g.aux != nullptr && g.ldaux > 1;
    if (MODE == MODE_NT) {
        if (splitk) {
            if ((int)blockIdx.y >= g.ldaux) return;
            const int chunk = ((Kred + g.ldaux - 1) / g.ldaux + BK - 1) / BK * BK;
            kbeg = blockIdx.y * chunk;
            kend = min(Kred, kbeg + chunk);
        } else if (blockIdx.y > 0) {
            return;
        }
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // a wave whose 32x32 sub-tile is entirely out of range skips the matrix work (wave-uniform)
    const bool wave_live = (m0 + wm * 32 < Mo) && (n0 + wn * 32 < No);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // Operand slabs travel global -> registers -> LDS.  The register stage is a 3-slot ring indexed statically: the loads
    // of slab s+2 are issued while slab s is multiplied, so a slab has two full iterations (~1 us of MFMA work) to arrive.
    // With a single slab in flight every iteration waited out most of a global round trip (measured ~2 us per 32-wide
    // slab on the [3200..5856] x [64..256] x [256..576] layers of the CNN / recurrent / MuJoCo-shaped networks).
    float4 ra[3][2], rb[3][2];
    __amdgpu_buffer_rsrc_t rsa, rsb;
    if (MODE == MODE_NT) { rsa = gemm_rsrc(g.A, g.M, g.lda, g.K); rsb = gemm_rsrc(g.B, g.N, g.ldb, g.K); }
    else if (MODE == MODE_NN) { rsa = gemm_rsrc(g.A, g.M, g.lda, g.K); rsb = gemm_rsrc(g.B, g.K, g.ldb, g.N); }
    else { rsa = gemm_rsrc(g.A, g.M, g.lda, g.N); rsb = gemm_rsrc(g.B, g.M, g.ldb, g.K); }
    // (reduction indices >= kend are out of range for the loaders: a slab past the end costs its load instructions, no traffic)
    auto load_tiles = [&](int k0, float4 (&qa)[2], float4 (&qb)[2]) {
        if (MODE == MODE_NT) {
            load_kcontig<VA>(rsa, g.lda, g.M, kend, m0, k0, qa);
            load_kcontig<VB>(rsb, g.ldb, g.N, kend, n0, k0, qb);
        } else if (MODE == MODE_NN) {
            load_kcontig<VA>(rsa, g.lda, g.M, kend, m0, k0, qa);
            load_kmajor<VB>(rsb, g.ldb, kend, g.N, k0, n0, -1, qb);
        } else {
            load_kmajor<VA>(rsa, g.lda, kend, g.N, k0, m0, -1, qa);                       // dY[m, n']
            load_kmajor<VB>(rsb, g.ldb, kend, g.K, k0, n0, g.dbias ? g.K : -1, qb);      // X[m, k'] (+ ones column)
        }
    };

    load_tiles(kbeg, ra[0], rb[0]);
    int k0 = kbeg;
#define GEMM_SLAB_STEP(CUR, KC)                                                                                          \
    {                                                                                                                    \
        if (MODE == MODE_TN) stage_kmajor(sA, ra[CUR]); else stage_kcontig(sA, ra[CUR]);                                 \
        if (MODE == MODE_NT) stage_kcontig(sB, rb[CUR]); else stage_kmajor(sB, rb[CUR]);                                 \
        lds_barrier();                             /* LDS-only: __syncthreads() would also drain the prefetches */       \
        if (wave_live && (KC) < kend) {            /* (a step past the end: staged zeros, no products) */                \
            _Pragma("unroll") for (int q = 0; q < BK / 8; ++q) {                                                         \
                float a4[4], b4[4];                                                                                      \
                if (MODE == MODE_TN) {                                                                                   \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s) a4[s] = sA[(q * 8 + 4 * lh + s) * LDR + wm * 32 + li]; \
                } else {                                                                                                 \
                    const float4 v = *reinterpret_cast<const float4*>(&sA[(wm * 32 + li) * LDK + q * 8 + 4 * lh]);       \
                    a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w;                                                  \
                }                                                                                                        \
                if (MODE == MODE_NT) {                                                                                   \
                    const float4 v = *reinterpret_cast<const float4*>(&sB[(wn * 32 + li) * LDK + q * 8 + 4 * lh]);       \
                    b4[0] = v.x; b4[1] = v.y; b4[2] = v.z; b4[3] = v.w;                                                  \
                } else {                                                                                                 \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s) b4[s] = sB[(q * 8 + 4 * lh + s) * LDR + wn * 32 + li]; \
                }                                                                                                        \
                _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                            \
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc, 0, 0, 0);                              \
            }                                                                                                            \
        }                                                                                                                \
        lds_barrier();                             /* slab consumed */                                                   \
    }
    /* Three slabs per trip.  The only loads in flight across the loop's back edge are those of slot 0 -- the first thing a trip
     * needs -- so the wait the compiler puts at the loop head costs nothing extra; slots 1 and 2 are requested at the top of the
     * trip, before slot 0 is waited for, and slot 0 of the NEXT trip right after its registers are staged.  (With loads of two
     * slots pending at the back edge the compiler waited for all of them at the head.) */
    while (k0 < kend) {
        load_tiles(k0 + BK, ra[1], rb[1]);
        load_tiles(k0 + 2 * BK, ra[2], rb[2]);
        GEMM_SLAB_STEP(0, k0)
        load_tiles(k0 + 3 * BK, ra[0], rb[0]);
        GEMM_SLAB_STEP(1, k0 + BK)
        GEMM_SLAB_STEP(2, k0 + 2 * BK)
        k0 += 3 * BK;
    }
#undef GEMM_SLAB_STEP

    if (!wave_live) return;
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int col = n0 + wn * 32 + li;
    if (col >= No) return;
    if (splitk) {                                  // raw partial sums of this K range; bias / activation in the epilogue
        float* ws = const_cast<float*>(g.aux) + (size_t)blockIdx.y * Mo * No;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < Mo) ws[(size_t)row * No + col] = acc[r];
        }
        return;
    }
    float bias = 0.f;
    if (MODE == MODE_NT && g.bias) bias = g.bias[col];
    if (MODE == MODE_NT) {
        XRL_ACT_DISPATCH(g.act,
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < Mo) g.C[(size_t)row * g.ldc + col] = act_apply_c<ACT>(acc[r] + bias);
            })
    } else if (MODE == MODE_NN) {
        if (g.aux) {
            XRL_ACT_DISPATCH(g.act,
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < Mo) g.C[(size_t)row * g.ldc + col] = acc[r] * act_grad_c<ACT>(g.aux[(size_t)row * g.ldaux + col]);
                })
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < Mo) g.C[(size_t)row * g.ldc + col] = acc[r];
            }
        }
    } else {
        const size_t so = (size_t)blockIdx.y * p.slab_stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= Mo) continue;
            if (col < g.K) g.C[so + (size_t)row * g.ldc + col] = acc[r];
            else g.dbias[so + row] = acc[r];          // the ones column: sum_m dY[m, row]
        }
    }
}

// C[m, n] = act(sum_s ws[s][m][n] + bias[n]) of the split-K groups of a forward launch (fixed summation order)
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(GemmBatch p) {
    const xrl_gemm_t& g = p.g[blockIdx.y];
    if (!(g.aux && g.ldaux > 1)) return;
    const int64_t total = (int64_t)g.M * g.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / g.N), n = (int)(i - (int64_t)m * g.N);
        float s = g.aux[i];
        int q = 1;
        for (; q + 8 <= g.ldaux; q += 8) {              // (eight partials requested at a time, added in order: one load per trip was a
            float w8[8];                                //  chain of ldaux global round trips)
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = g.aux[(size_t)(q + j) * total + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += w8[j];
        }
        for (; q < g.ldaux; ++q) s += g.aux[(size_t)q * total + i];
        const float z = s + (g.bias ? g.bias[n] : 0.f);
        float y = z;
        XRL_ACT_DISPATCH(g.act, y = act_apply_c<ACT>(z);)
        g.C[(size_t)m * g.ldc + n] = y;
    }
}

// ---- skinny layers -------------------------------------------------------------------------------------------------
// A layer with a handful of outputs and a long reduction (the 512 -> n_actions output layer of the Atari Q head on a
// batch of 32..96 rows) is ONE 64x64 tile for the kernel above: a single workgroup walking K = 512 in 16 dependent
// global->LDS->MFMA rounds (measured 27.7 us, forward and data-gradient each).  These two kernels spread the same layer
// over one wave per row (forward) / one thread per input-gradient element (backward): ~3 us.
constexpr int SKINNY_N = 16;

__global__ void __launch_bounds__(64) skinny_fwd_kernel(GemmBatch p) {
    const xrl_gemm_t& g = p.g[blockIdx.y];
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= g.M) return;
    float acc[SKINNY_N];
#pragma unroll
    for (int n = 0; n < SKINNY_N; ++n) acc[n] = 0.f;
    const float* __restrict__ a = g.A + (size_t)m * g.lda;
    const float* __restrict__ B = g.B;
    const int N = g.N;
    int k = lane;
    for (; k + 64 * 3 < g.K; k += 64 * 4) {              // 4 k-steps: every load issued before the first FMA
        float av[4], bv[4][SKINNY_N];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = a[k + 64 * u];
#pragma unroll
            for (int n = 0; n < SKINNY_N; ++n) bv[u][n] = n < N ? B[(size_t)n * g.ldb + k + 64 * u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int n = 0; n < SKINNY_N; ++n) acc[n] = fmaf(av[u], bv[u][n], acc[n]);
    }
    for (; k < g.K; k += 64) {
        const float av = a[k];
#pragma unroll
        for (int n = 0; n < SKINNY_N; ++n)
            if (n < N) acc[n] = fmaf(av, B[(size_t)n * g.ldb + k], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < SKINNY_N; ++n) {
        if (n >= g.N) break;
        const float s = wave_sum(acc[n]);
        if (lane == 0) {
            const float z = s + (g.bias ? g.bias[n] : 0.f);
            float y = z;
            XRL_ACT_DISPATCH(g.act, y = act_apply_c<ACT>(z);)
            g.C[(size_t)m * g.ldc + n] = y;
        }
    }
}

// C[m, c] = (sum_{k < K} A[m, k] B[k, c]) * act'(aux[m, c]),  K <= SKINNY_N
__global__ void __launch_bounds__(256) skinny_bwd_data_kernel(GemmBatch p) {
    const xrl_gemm_t& g = p.g[blockIdx.y];
    const int64_t total = (int64_t)g.M * g.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / g.N), c = (int)(i - (int64_t)m * g.N);
        float s = 0.f;
        for (int k = 0; k < g.K; ++k) s = fmaf(g.A[(size_t)m * g.lda + k], g.B[(size_t)k * g.ldb + c], s);
        if (g.aux) {
            const float y = g.aux[(size_t)m * g.ldaux + c];
            XRL_ACT_DISPATCH(g.act, s *= act_grad_c<ACT>(y);)
        }
        g.C[(size_t)m * g.ldc + c] = s;
    }
}

// Data gradient of a few-row layer with a LONG reduction (dX[32, 64] = dY[32, 512] . W[512, 64] of the Atari head: one
// 64x64 tile walking K = 512, measured 26 us): one wave per (row, 64 output columns), the row's dY values broadcast, W
// rows read coalesced.
__global__ void __launch_bounds__(256) rowwise_bwd_data_kernel(GemmBatch p) {
    __shared__ float part[4][64];
    const xrl_gemm_t& g = p.g[blockIdx.z];
    const int m = blockIdx.x, lane = threadIdx.x & 63, kq = threadIdx.x >> 6;     // 4 waves split the reduction
    const int c = blockIdx.y * 64 + lane;
    const bool live = m < g.M && c < g.N;
    const float* __restrict__ a = g.A + (size_t)(live ? m : 0) * g.lda;
    const float* __restrict__ bcol = g.B + (live ? c : 0);
    float s = 0.f;
    constexpr int U = 8;                                   // loads of U steps are issued before the first FMA
    int k = kq;
    for (; k + 4 * (U - 1) < g.K; k += 4 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { av[u] = a[k + 4 * u]; bv[u] = bcol[(size_t)(k + 4 * u) * g.ldb]; }
#pragma unroll
        for (int u = 0; u < U; ++u) s = fmaf(av[u], bv[u], s);
    }
    for (; k < g.K; k += 4) s = fmaf(a[k], bcol[(size_t)k * g.ldb], s);
    part[kq][lane] = s;
    __syncthreads();
    if (kq == 0 && live) {
        float r = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (g.aux) {
            const float y = g.aux[(size_t)m * g.ldaux + c];
            XRL_ACT_DISPATCH(g.act, r *= act_grad_c<ACT>(y);)
        }
        g.C[(size_t)m * g.ldc + c] = r;
    }
}

static bool all_rowwise_nn(int mode, const xrl_gemm_t* groups, int n_groups) {
    if (mode != MODE_NN) return false;
    for (int i = 0; i < n_groups; ++i)
        if (!(groups[i].M <= 128 && groups[i].K >= 512)) return false;
    return true;
}

static bool all_skinny(int mode, const xrl_gemm_t* groups, int n_groups) {
    for (int i = 0; i < n_groups; ++i) {
        const xrl_gemm_t& g = groups[i];
        // few rows only: with thousands of rows the tiled kernel already fills the chip (and the PPO / QMIX paths, whose
        // results are pinned bit for bit between their kernel variants, never come here)
        if (g.M > 256) return false;
        if (mode == MODE_NT && !(g.N <= SKINNY_N && g.K >= 512)) return false;
        if (mode == MODE_NN && !(g.K <= SKINNY_N && g.N >= 512)) return false;
    }
    return mode == MODE_NT || mode == MODE_NN;
}

static int launch(int mode, const xrl_gemm_t* groups, int n_groups, int n_split, int64_t slab_stride,
                  xrl_stream_t stream, bool skip_epilogue = false) {
    XRL_CHECK_ARG(groups != nullptr && n_groups >= 1 && n_groups <= MAX_GROUPS);
    XRL_CHECK_ARG(n_split >= 1 && n_split <= 65535);
    if (all_rowwise_nn(mode, groups, n_groups)) {
        GemmBatch b;
        b.n_groups = n_groups; b.n_split = 1; b.slab_stride = 0;
        int maxM = 0, maxN = 0;
        for (int i = 0; i < n_groups; ++i) {
            XRL_CHECK_ARG(groups[i].A && groups[i].B && groups[i].C && groups[i].M > 0 && groups[i].N > 0 && groups[i].K > 0);
            b.g[i] = groups[i];
            maxM = groups[i].M > maxM ? groups[i].M : maxM;
            maxN = groups[i].N > maxN ? groups[i].N : maxN;
        }
        hipLaunchKernelGGL(rowwise_bwd_data_kernel, dim3(maxM, (maxN + 63) / 64, n_groups), dim3(256), 0, as_stream(stream), b);
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    if (all_skinny(mode, groups, n_groups)) {
        GemmBatch b;
        b.n_groups = n_groups; b.n_split = 1; b.slab_stride = 0;
        int maxM = 0; int64_t max_el = 0;
        for (int i = 0; i < n_groups; ++i) {
            XRL_CHECK_ARG(groups[i].A && groups[i].B && groups[i].C && groups[i].M > 0 && groups[i].N > 0 && groups[i].K > 0);
            b.g[i] = groups[i];
            maxM = groups[i].M > maxM ? groups[i].M : maxM;
            const int64_t el = (int64_t)groups[i].M * groups[i].N;
            max_el = el > max_el ? el : max_el;
        }
        if (mode == MODE_NT)
            hipLaunchKernelGGL(skinny_fwd_kernel, dim3(maxM, n_groups), dim3(64), 0, as_stream(stream), b);
        else
            hipLaunchKernelGGL(skinny_bwd_data_kernel, dim3((unsigned)((max_el + 255) / 256), n_groups), dim3(256), 0,
                               as_stream(stream), b);
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    GemmBatch b;
    b.n_groups = n_groups; b.n_split = n_split; b.slab_stride = slab_stride;
    int max_tiles = 0;
    for (int i = 0; i < n_groups; ++i) {
        const xrl_gemm_t& g = groups[i];
        XRL_CHECK_ARG(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0);
        XRL_CHECK_ARG(g.lda > 0 && g.ldb > 0 && g.ldc > 0);
        b.g[i] = g;
        int Mo, No;
        if (mode == MODE_TN) { Mo = g.N; No = g.K + (g.dbias ? 1 : 0); } else { Mo = g.M; No = g.N; }
        const int tiles = ((Mo + BM - 1) / BM) * ((No + BN - 1) / BN);
        max_tiles = tiles > max_tiles ? tiles : max_tiles;
    }
    int ksplit = 1;                                // forward split-K: groups that carry a workspace in `aux`
    int64_t max_el = 0;
    if (mode == MODE_NT)
        for (int i = 0; i < n_groups; ++i)
            if (groups[i].aux && groups[i].ldaux > 1) {
                XRL_CHECK_ARG(groups[i].ldaux <= 64);
                ksplit = groups[i].ldaux > ksplit ? groups[i].ldaux : ksplit;
                const int64_t el = (int64_t)groups[i].M * groups[i].N;
                max_el = el > max_el ? el : max_el;
            }
    dim3 grid(max_tiles, mode == MODE_TN ? n_split : ksplit, n_groups);
    // 16-byte operand loads need quads that are entirely inside or outside the matrix: leading dimension and the extent along
    // the contiguous direction multiples of 4, base 16-byte aligned -- for every group of the launch (else 4-byte loads)
    bool va = true, vb = true;
    for (int i = 0; i < n_groups; ++i) {
        const xrl_gemm_t& g = groups[i];
        const int ea = mode == MODE_TN ? g.N : g.K, eb = mode == MODE_NT ? g.K : (mode == MODE_NN ? g.N : g.K);
        const int rows_a = g.M, rows_b = mode == MODE_NT ? g.N : (mode == MODE_NN ? g.K : g.M);
        va = va && (g.lda % 4 == 0) && (ea % 4 == 0) && (reinterpret_cast<uintptr_t>(g.A) % 16 == 0);
        vb = vb && (g.ldb % 4 == 0) && (eb % 4 == 0) && (reinterpret_cast<uintptr_t>(g.B) % 16 == 0);
        XRL_CHECK_ARG((int64_t)rows_a * g.lda * 4 < ((int64_t)1 << 31) && (int64_t)rows_b * g.ldb * 4 < ((int64_t)1 << 31));
        XRL_CHECK_ARG(g.lda < (1 << 24) && g.ldb < (1 << 24) && rows_a < (1 << 24) && rows_b < (1 << 24));
    }
#define GEMM_LAUNCH(M_)                                                                                                  \
    {                                                                                                                    \
        if (va && vb) hipLaunchKernelGGL((gemm_f32_kernel<M_, true, true>), grid, 256, 0, as_stream(stream), b);          \
        else if (va) hipLaunchKernelGGL((gemm_f32_kernel<M_, true, false>), grid, 256, 0, as_stream(stream), b);          \
        else if (vb) hipLaunchKernelGGL((gemm_f32_kernel<M_, false, true>), grid, 256, 0, as_stream(stream), b);          \
        else hipLaunchKernelGGL((gemm_f32_kernel<M_, false, false>), grid, 256, 0, as_stream(stream), b);                 \
    }
    if (mode == MODE_NT && ksplit > 1) {
        GEMM_LAUNCH(MODE_NT)
        if (skip_epilogue) { XRL_CHECK_LAUNCH(); return XRL_OK; }       // (xrl_linear_fwd_partials: the caller's launch sums ws[s][m][n])
        unsigned nb = (unsigned)((max_el + 255) / 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(nb, n_groups), dim3(256), 0, as_stream(stream), b);
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    if (mode == MODE_NT) GEMM_LAUNCH(MODE_NT)
    else if (mode == MODE_NN) GEMM_LAUNCH(MODE_NN)
    else GEMM_LAUNCH(MODE_TN)
#undef GEMM_LAUNCH
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

}  // namespace xrl

extern "C" int xrl_linear_fwd(const xrl_gemm_t* groups, int n_groups, xrl_stream_t stream) {
    return xrl::launch(xrl::MODE_NT, groups, n_groups, 1, 0, stream);
}
extern "C" int xrl_linear_fwd_partials(const xrl_gemm_t* groups, int n_groups, xrl_stream_t stream) {
    XRL_CHECK_ARG(groups && n_groups >= 1);
    for (int i = 0; i < n_groups; ++i) XRL_CHECK_ARG(groups[i].aux && groups[i].ldaux > 1);       // every group leaves partials
    return xrl::launch(xrl::MODE_NT, groups, n_groups, 1, 0, stream, true);
}
extern "C" int xrl_linear_bwd_data(const xrl_gemm_t* groups, int n_groups, xrl_stream_t stream) {
    return xrl::launch(xrl::MODE_NN, groups, n_groups, 1, 0, stream);
}
extern "C" int xrl_linear_bwd_weight(const xrl_gemm_t* groups, int n_groups, int n_split, int64_t slab_stride,
                                     xrl_stream_t stream) {
    return xrl::launch(xrl::MODE_TN, groups, n_groups, n_split, slab_stride, stream);
}
