// Whole feed-forward plans as ONE launch (xrl_mlp_chain_fwd, include/xrl_hip.h): a plan -- mlp_block chains with stacked / parallel
// layers (layers.py:16-33; the actor-critic of configs/ppo/classic_control/*.yaml, the fc -> W_ih chain below a GRU and the Q head above
// it, rnn.py:52-77, the mixer's hyper-networks, mixers.py) -- was one grouped GEMM launch per STAGE, 6-8 us each for a few hundred
// kiloflops; the stages of a plan are a dependent chain, so a 3-stage acting pass was three launches one after the other.  Here a
// workgroup of 8 waves takes a 32-row tile of ONE job (up to four independent jobs per launch: an eval network, its target twin, the
// two mixers), keeps every level of the tile in LDS and walks the job's layers with mlp_tile.h's fused_layer -- the MFMA tile of the
// one-launch rollout / minibatch kernels -- in its 1 : 1 wave-to-column-tile form: every output is the same k-ordered fma chain as
// gemm.hip's (K in chunks of 8: k = 8 q + 4 h + s), so the launch is BIT-IDENTICAL to the per-stage launches (tests/test_gpu_primitives.py).
// Levels the caller asks for (all of them when a backward pass follows) are written back coalesced.
#include "common.h"
#include "mlp_tile.h"

namespace xrl {

__device__ long long* g_mc_dbg = nullptr;      // tools/probe_mlp_chain.py: clock stamps of workgroup 0 (xrl_debug_mlp_chain_stamps)
#define MCSTAMP() do { if (mdbg && threadIdx.x == 0 && blockIdx.x == 0 && mdi < 15) mdbg[mdi++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

// per-layer constants, formed once per workgroup (thread l < n_layers) and read from LDS afterwards: recomputing LDS offsets from the
// kernel arguments wherever they were needed -- loops of dependent scalar loads, each a scalar-cache miss on first touch -- made
// every phase of the launch 4-6 us (stamps: staging 11.5 k cycles, the 6-wide first layer 13.6 k, write-back 8 k)
enum { MC_IN = 0, MC_LDIN, MC_OUT, MC_LDOUT, MC_BIAS, MC_N, MC_K, MC_ACT, MC_WOFF, MC_T0, MC_STAGE_END, MC_OUTLVL, MC_OUTOFF, MC_BOFF, MC_W = 16 };

__global__ void __launch_bounds__(FUSED_THREADS) mlp_chain_kernel(xrl_mlp_chain_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int tab[XRL_FUSED_MAX_LAYERS][MC_W];
    __shared__ int s_lvl_off[XRL_FUSED_MAX_LEVELS + 1], s_lvl_ld[XRL_FUSED_MAX_LEVELS];
    const int tid = threadIdx.x;
    long long* const mdbg = g_mc_dbg;
    int mdi = 0;
    MCSTAMP();
    int j = 0;
#pragma unroll
    for (int q = 1; q < XRL_CHAIN_MAX_JOBS; ++q) if (q < p.n_jobs && (int)blockIdx.x >= p.tile0[q]) j = q;
    const xrl_mlp_chain_job_t& J = p.job[j];
    const int tile = (int)blockIdx.x - p.tile0[j], r0 = tile * FT, rows = min(FT, J.M - r0);
    const int nL = J.n_layers, nLv = J.n_levels;
    const float* const params = J.params;
    // ---- the table: level offsets (thread 64), per-layer constants (threads 0..7)
    if (tid == 64) {
        int o = 0;
        for (int l = 0; l < nLv; ++l) { s_lvl_off[l] = o; s_lvl_ld[l] = level_ld(J.level_width[l]); o += FT * level_ld(J.level_width[l]); }
        s_lvl_off[nLv] = o;
    }
    if (tid < nL) {
        const xrl_fused_layer_t& L = J.layers[tid];
        int lo_in = 0, lo_out = 0, acts = 0, cb = 0, t0 = 0, f = 0;
        for (int l = 0; l < nLv; ++l) {
            if (l < L.in_level) lo_in += FT * level_ld(J.level_width[l]);
            if (l < L.out_level) lo_out += FT * level_ld(J.level_width[l]);
            acts += FT * level_ld(J.level_width[l]);
        }
        for (int m = 0; m < tid; ++m) cb += (J.layers[m].N + 3) / 4 * 4;
        for (int m = 0; m < tid; ++m) { if (J.layers[m].out_level != L.out_level) { f = m + 1; } }
        for (int m = f; m < tid; ++m) t0 += (J.layers[m].N + 31) / 32;           // first tile of this layer inside its stage
        int e = tid + 1;
        while (e < nL && J.layers[e].out_level == L.out_level) ++e;
        int* T = tab[tid];
        T[MC_IN] = lo_in + L.in_off; T[MC_LDIN] = level_ld(J.level_width[L.in_level]);
        T[MC_OUT] = lo_out + L.out_off; T[MC_LDOUT] = level_ld(J.level_width[L.out_level]);
        T[MC_BIAS] = acts + cb; T[MC_N] = L.N; T[MC_K] = L.K; T[MC_ACT] = L.act; T[MC_WOFF] = L.w_off; T[MC_T0] = t0;
        T[MC_STAGE_END] = e; T[MC_OUTLVL] = L.out_level; T[MC_OUTOFF] = L.out_off; T[MC_BOFF] = L.b_off;
    }
    // ---- input rows (loads first)
    const int K0 = J.level_width[0], n_in = rows * K0, ldx = J.ldx;
    const float* const xg = J.x;
    float xin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + u * FUSED_THREADS;
        xin[u] = 0.f;
        if (i < n_in) { const int r = i / K0, k = i - r * K0; xin[u] = xg[(size_t)(r0 + r) * ldx + k]; }
    }
    __syncthreads();
    const int acts_end = s_lvl_off[nLv];
    const int total = tab[nL - 1][MC_BIAS] + (tab[nL - 1][MC_N] + 3) / 4 * 4;
    // (biases: value u of this thread is element tid + u * 512 of the layers' bias vectors laid end to end)
    float bin[2];
    int bdst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int i = tid + u * FUSED_THREADS, l = 0;
        while (l < nL && i >= tab[l][MC_N]) { i -= tab[l][MC_N]; ++l; }
        bin[u] = 0.f; bdst[u] = -1;
        if (l < nL) { bin[u] = params[tab[l][MC_BOFF] + i]; bdst[u] = tab[l][MC_BIAS] + i; }
    }
    for (int i = tid; i < total; i += FUSED_THREADS) lds[i] = 0.f;          // padding columns / rows must read as zero
    __syncthreads();
    {
        const int ld0 = s_lvl_ld[0];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * FUSED_THREADS;
            if (i < n_in) { const int r = i / K0, k = i - r * K0; lds[r * ld0 + k] = xin[u]; }
        }
        for (int i = tid + 4 * FUSED_THREADS; i < n_in; i += FUSED_THREADS) {        // (wider inputs than 64 columns: plain tail)
            const int r = i / K0, k = i - r * K0;
            lds[r * ld0 + k] = xg[(size_t)(r0 + r) * ldx + k];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) if (bdst[u] >= 0) lds[bdst[u]] = bin[u];
    {                                                                                  // (more than 1 024 bias values: plain tail)
        int run = 0;
        for (int l = 0; l < nL; ++l) {
            const int N = tab[l][MC_N], cb = tab[l][MC_BIAS];
            for (int i = tid + 2 * FUSED_THREADS - run; i < N; i += FUSED_THREADS)
                if (i >= 0) lds[cb + i] = params[tab[l][MC_BOFF] + i];
            run += N;
        }
    }
    __syncthreads();
    MCSTAMP();
    // ---- the stages: the layers of a stage (consecutive layers with the same output level; they read levels finished before the
    //      stage) share the eight waves -- wave w takes column tile w, w + 8, ... of the stage's tiles laid end to end -- and the
    //      weights of the NEXT stage's first pass are requested before this stage multiplies (16 chunks of 8 k per wave in registers:
    //      K <= 128), so a stage costs its MFMA chain, not a global round trip + the chain.
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    auto tile_of = [&](int f, int e, int t, int& l, int& tl) {        // tile t of stage [f, e) -> layer l, its column tile tl
        l = -1; tl = 0;
        for (int m = f; m < e; ++m) {
            const int nt = (tab[m][MC_N] + 31) / 32, t0 = tab[m][MC_T0];
            if (t >= t0 && t < t0 + nt) { l = m; tl = t - t0; }
        }
        l = __builtin_amdgcn_readfirstlane(l); tl = __builtin_amdgcn_readfirstlane(tl);
    };
    auto fetch = [&](int f, int e, float4 (&pf)[PD]) -> bool {           // this wave's first-pass weights of stage [f, e)
        int l, tl;
        tile_of(f, e, wave, l, tl);
        if (l < 0) return false;
        const int K = __builtin_amdgcn_readfirstlane(tab[l][MC_K]), N = __builtin_amdgcn_readfirstlane(tab[l][MC_N]);
        const float* Wg = params + __builtin_amdgcn_readfirstlane(tab[l][MC_WOFF]);
        if ((K & 7) || K > 8 * PD || (reinterpret_cast<uintptr_t>(Wg) & 15)) return false;
        const float* wrow = Wg + (size_t)min(tl * 32 + li, N - 1) * K + 4 * lh;
        const int kq = K >> 3;
#pragma unroll
        for (int q = 0; q < PD; ++q) pf[q] = q < kq ? *reinterpret_cast<const float4*>(wrow + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
        return true;
    };
    auto run_stage = [&](int f, int e, const float4 (&pf)[PD], bool have_pf) {
        const int T = __builtin_amdgcn_readfirstlane(tab[e - 1][MC_T0] + (tab[e - 1][MC_N] + 31) / 32);
        for (int t0 = 0; t0 < T; t0 += NW) {
            int l, tl;
            tile_of(f, e, t0 + wave, l, tl);
            if (l < 0) continue;
            const int K = __builtin_amdgcn_readfirstlane(tab[l][MC_K]), N = __builtin_amdgcn_readfirstlane(tab[l][MC_N]);
            const int kq = (K + 7) / 8, n0 = tl * 32, wr = min(n0 + li, N - 1);
            const int ld_in = __builtin_amdgcn_readfirstlane(tab[l][MC_LDIN]);
            const float* arow = lds + __builtin_amdgcn_readfirstlane(tab[l][MC_IN]) + li * ld_in + 4 * lh;
            const float* Wg = params + __builtin_amdgcn_readfirstlane(tab[l][MC_WOFF]);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            if (have_pf && t0 == 0) {
                float4 af[PD];                                          // all A fragments first: one LDS latency, then pure MFMA
#pragma unroll
                for (int q = 0; q < PD; ++q) af[q] = q < kq ? *reinterpret_cast<const float4*>(arow + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < PD; ++q) if (q < kq) { MFMA4(af[q], pf[q], acc) }
            } else if (((K & 7) == 0) && ((reinterpret_cast<uintptr_t>(Wg) & 15) == 0)) {
                const float* wrow = Wg + (size_t)wr * K + 4 * lh;
#pragma unroll 8
                for (int q = 0; q < kq; ++q) {
                    const float4 bq = *reinterpret_cast<const float4*>(wrow + q * 8);
                    const float4 aq = *reinterpret_cast<const float4*>(arow + q * 8);
                    MFMA4(aq, bq, acc)
                }
            } else {
                const float* wrow = Wg + (size_t)wr * K;
                for (int q = 0; q < kq; ++q) {
                    const int kk = q * 8 + 4 * lh;
                    const float4 aq = *reinterpret_cast<const float4*>(arow + q * 8);
                    float4 bq;
                    bq.x = kk + 0 < K ? wrow[kk + 0] : 0.f;
                    bq.y = kk + 1 < K ? wrow[kk + 1] : 0.f;
                    bq.z = kk + 2 < K ? wrow[kk + 2] : 0.f;
                    bq.w = kk + 3 < K ? wrow[kk + 3] : 0.f;
                    MFMA4(aq, bq, acc)
                }
            }
            // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            const int col = n0 + li;
            if (col < N) {
                float* out = lds + __builtin_amdgcn_readfirstlane(tab[l][MC_OUT]);
                const int ld_out = __builtin_amdgcn_readfirstlane(tab[l][MC_LDOUT]);
                const float bv = lds[__builtin_amdgcn_readfirstlane(tab[l][MC_BIAS]) + col];
                XRL_ACT_DISPATCH(__builtin_amdgcn_readfirstlane(tab[l][MC_ACT]),
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        out[row * ld_out + col] = act_apply_c<ACT>(acc[r] + bv);
                    })
            }
        }
        __syncthreads();
    };
    {
        float4 pfa[PD], pfb[PD];
        int f = 0;
        bool have_a = fetch(0, __builtin_amdgcn_readfirstlane(tab[0][MC_STAGE_END]), pfa), have_b = false;
        // (two stages per trip so that the two register sets alternate without copies)
        while (f < nL) {
            const int e = __builtin_amdgcn_readfirstlane(tab[f][MC_STAGE_END]);
            if (e < nL) have_b = fetch(e, __builtin_amdgcn_readfirstlane(tab[e][MC_STAGE_END]), pfb);
            run_stage(f, e, pfa, have_a);
            MCSTAMP();
            f = e;
            if (f >= nL) break;
            const int e2 = __builtin_amdgcn_readfirstlane(tab[f][MC_STAGE_END]);
            if (e2 < nL) have_a = fetch(e2, __builtin_amdgcn_readfirstlane(tab[e2][MC_STAGE_END]), pfa);
            run_stage(f, e2, pfb, have_b);
            MCSTAMP();
            f = e2;
        }
    }
    // ---- write-back: every layer's block of a requested level (columns no layer writes stay as the caller left them)
#pragma unroll 1
    for (int l = 0; l < nL; ++l) {
        const int lvl = tab[l][MC_OUTLVL];
        float* dst = J.out[lvl];
        if (!dst) continue;
        const int W = tab[l][MC_N], ld = J.ld_out[lvl], n = rows * W, lo = tab[l][MC_OUT], ll = tab[l][MC_LDOUT], oo = tab[l][MC_OUTOFF];
        for (int i0 = tid; i0 < n; i0 += 4 * FUSED_THREADS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * FUSED_THREADS;
                if (i < n) { const int r = i / W, k = i - r * W; dst[(size_t)(r0 + r) * ld + oo + k] = lds[lo + r * ll + k]; }
            }
        }
    }
    MCSTAMP();
    if (mdbg && threadIdx.x == 0 && blockIdx.x == 0) mdbg[15] = mdi;
}
#undef MCSTAMP

static size_t chain_lds_floats(const xrl_mlp_chain_job_t& J) {
    size_t off = 0;
    for (int l = 0; l < J.n_levels; ++l) off += (size_t)FT * level_ld(J.level_width[l]);
    for (int l = 0; l < J.n_layers; ++l) off += (size_t)(J.layers[l].N + 3) / 4 * 4;
    return off;
}

}  // namespace xrl

using namespace xrl;

static int chain_check(const xrl_mlp_chain_t& p, size_t& bytes) {
    XRL_CHECK_ARG(p.n_jobs >= 1 && p.n_jobs <= XRL_CHAIN_MAX_JOBS && p.tile0[0] == 0);
    bytes = 0;
    for (int j = 0; j < p.n_jobs; ++j) {
        const xrl_mlp_chain_job_t& J = p.job[j];
        XRL_CHECK_ARG(J.x && J.params && J.M > 0 && J.n_layers >= 1 && J.n_layers <= XRL_FUSED_MAX_LAYERS && J.n_levels >= 2 &&
                      J.n_levels <= XRL_FUSED_MAX_LEVELS && J.ldx >= J.level_width[0] && J.level_width[0] >= 1);
        XRL_CHECK_ARG(p.tile0[j + 1] - p.tile0[j] == (J.M + FT - 1) / FT);
        for (int l = 0; l < J.n_layers; ++l) {
            const xrl_fused_layer_t& L = J.layers[l];
            XRL_CHECK_ARG(L.K >= 1 && L.N >= 1 && L.in_level >= 0 && L.in_level < L.out_level && L.out_level < J.n_levels &&
                          L.in_off >= 0 && L.in_off + L.K <= J.level_width[L.in_level] && L.out_off >= 0 &&
                          L.out_off + L.N <= J.level_width[L.out_level]);
            XRL_CHECK_ARG((L.in_off & 3) == 0);                               // (16-byte LDS reads of the A fragments)
        }
        for (int l = 1; l < J.n_levels; ++l) XRL_CHECK_ARG(J.out[l] == nullptr || J.ld_out[l] >= J.level_width[l]);
        const size_t b = chain_lds_floats(J) * 4;
        bytes = b > bytes ? b : bytes;
    }
    return XRL_OK;
}

extern "C" int xrl_debug_mlp_chain_stamps(long long* stamps) {
    XRL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mc_dbg), &stamps, sizeof(stamps)));
    return XRL_OK;
}

extern "C" int xrl_mlp_chain_lds_bytes(const xrl_mlp_chain_t* p) {
    if (!p) return -1;
    size_t bytes = 0;
    if (chain_check(*p, bytes) != XRL_OK) return -1;
    return (int)bytes;
}

extern "C" int xrl_mlp_chain_fwd(const xrl_mlp_chain_t* pp, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_mlp_chain_t& p = *pp;
    size_t bytes = 0;
    if (chain_check(p, bytes) != XRL_OK) return XRL_EINVAL;
    XRL_CHECK_ARG(bytes <= 160 * 1024);
    static size_t allowed = 0;
    if (bytes > allowed) {
        XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        allowed = bytes;
    }
    hipLaunchKernelGGL(mlp_chain_kernel, dim3(p.tile0[p.n_jobs]), dim3(FUSED_THREADS), bytes, as_stream(stream), p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
