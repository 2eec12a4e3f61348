// The headline's minibatch launch with its three 128-wide products as EXACT 3-WAY bf16 SPLITS on the matrix cores.
//
// csrc/ppo_trunk.hip spends 51 % of a wave's life behind v_mfma_f32_32x32x2_f32 (64 pipe cycles for 4 096 flop: the fp32 matrix
// instruction runs at the fp32 VECTOR rate and shares its issue with it -- DESIGN.md section 3 "Round 6" (b)).  The bf16 instruction
// v_mfma_f32_32x32x16_bf16 does 8x the k-depth in HALF the pipe cycles.  A float32 x is the exact sum of three bf16 numbers
//      h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round-to-nearest-even; 3 x 8 significand bits = 24; both
//                                                                  subtractions and the last conversion are exact)
// so a product of two float32 numbers is the exact sum of nine bf16 x bf16 products, each of which the matrix core forms exactly
// and accumulates in float32.  This kernel issues the six largest -- (h,h) (h,m) (m,h) (m,m) (h,l) (l,h): what is dropped,
// (m,l) + (l,m) + (l,l), is <= 2^-23 |x y| per scalar product, signed at random: the size of ONE float32 rounding of that product,
// i.e. of the error the fp32 instruction's own accumulation makes per term.  Six bf16 instructions (192 pipe cycles) replace eight fp32
// ones (512) for the same 32 x 32 x 16 block: the matrix pipe is busy 9.3 k instead of 24.8 k cycles per SIMD.
// What it is NOT: a bf16 computation.  No operand is rounded to 8 (or 16) bits; the parity fixtures hold at the same 1e-5 as the
// fp32 kernel (tests/test_gpu_ppo.py: test_split_product_minibatch_*), and against the float64 twin the gradients are as close as
// the fp32 kernel's (profiles/r06_q_*).
//
// Where the splits are made (a split costs ~4.5 vector instructions per element, and the vector ALU shares the matrix pipe's issue:
// splitting at every CONSUMER would cost what the bf16 instructions save):
//   * W1 (both orientations): by the OPTIMISER launch -- the fragment image this kernel streams holds three bf16 planes
//     (xrl_pack_mid_frags16; kept current through split mirror maps, xrl_mirrors_t.split_plane);
//   * h1 and g2 = dLoss/dz2: once, by the thread that PRODUCES the element, into three row-major 16-bit planes in LDS (6 bytes per
//     element instead of 4); all three products read them: forward and backward-data as 16-byte row reads, the weight gradient
//     (k = the row axis: a column slice per lane) through ds_read_b64_tr_b16, the LDS transpose read of gfx950;
//   * h2 stays float32 (only the vector phases read it).
// The first layer (k = 4: two fp32 instructions) stays on v_mfma_f32_32x32x2_f32: same numbers as ppo_trunk_kernel.  It and the
// branch layer are issued TRANSPOSED (A = weights, B = rows): a lane then holds 4 CONSECUTIVE columns of one row, and h1's planes /
// h2 are written as 8- / 16-byte stores instead of 2- / 4-byte ones.
//
// Class: the members of the shared-trunk family with D <= 8, A <= 4 on 64-row tiles -- the CartPole class (4, 2), the headline, as a
// compile-time instance (LDS 153 KB of the CU's 160); Acrobot (6, 3), LunarLander (8, 4), MountainCar (2, 3) and the Gaussian Pendulum
// (3, 1; tanh on the mean) through any-(D, A) instances (156 KB): every classic-control PPO yaml of the reference.  Wider rows
// (BipedalWalker's 24) do not fit beside the six planes and keep csrc/ppo_trunk.hip.  Everything outside the three products -- gather, head, loss, the small gradients, slab
// layout, loss partials -- is ppo_trunk_kernel<ACT, 0, 64, DS, AS>'s code, statement for statement.
// Reference semantics: memory_tools.py:267-287 (sample) + ppo_learner.py:46-62 (forward / loss / backward).
#include "common.h"
#include "mlp_tile.h"
#include "ppo_math.h"
#include "split3.h"
#include "sample.h"

namespace xrl {

typedef unsigned bu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bu32x2 __attribute__((ext_vector_type(2)));
typedef short bs16x4 __attribute__((ext_vector_type(4)));

constexpr int BH = 128;                      // hidden width (trunk and each branch)
constexpr int BLD = BH + 4;                  // float32 row stride (h2)
constexpr int BPL = BH + 8;                  // row stride of a 16-bit plane (elements; 272 bytes: 16-byte row reads conflict-free)
constexpr int BPT = 64;                      // rows per workgroup
constexpr int BPLANE = BPT * BPL;            // elements per plane
constexpr int BDMAX = 8, BAMAX = 4;          // observation / head width limits of the any-(D, A) instances (six planes + these rows: 160 KB)
constexpr int BXLD = 8;                      // row stride of the gathered observations

template <int DM, int AM>                    // rows for DM observation dims / AM head rows
struct BxLds {                               // byte offsets
    static constexpr int H1P = 0, G2P = H1P + 3 * BPLANE * 2, H2 = G2P + 3 * BPLANE * 2, XS = H2 + BPT * BLD * 4,
                         RSC = XS + BPT * BXLD * 4, DZH = RSC + BPT * 12 * 4, W0T = DZH + BPT * 16 * 4, B0 = W0T + DM * BLD * 4,
                         BM = B0 + BH * 4, WH = BM + BH * 4, BHS = WH + AM * BLD * 4, LS = BHS + 8 * 4, SRC = LS + 8 * 4, RST = SRC + BPT * 4,
                         BGP = RST + BPT * 5 * 8, BYTES = BGP + 2 * BH * 8;
    static_assert(BYTES <= 160 * 1024, "LDS");
    static_assert((RST & 7) == 0 && (H2 & 15) == 0 && (G2P & 15) == 0 && (B0 & 15) == 0 && (WH & 15) == 0, "alignment");
    // the row blocks' first-layer partial sums [2][128][DM + 1] go where the row scalars, head gradients and first-layer weights were
    static_assert(DZH == RSC + BPT * 12 * 4 && W0T == DZH + BPT * 16 * 4 && 2 * BH * (DM + 1) * 4 <= B0 - RSC, "first-layer partial sums");
};

__device__ __forceinline__ float plane_value(const unsigned short* pl, int o) {          // the float32 an element was split from (exact)
    return (bf16_bits_to_float(pl[o]) + bf16_bits_to_float(pl[BPLANE + o])) + bf16_bits_to_float(pl[2 * BPLANE + o]);
}

#define BX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0)
// acc += the six largest of the nine (part of x) x (part of y) products, smallest first; x = A-operand planes, y = B-operand planes
#define BX_MFMA6(xh, xm, xl, yh, ym, yl, acc)                                                  \
    acc = BX_MFMA(xl, yh, acc); acc = BX_MFMA(xh, yl, acc); acc = BX_MFMA(xm, ym, acc);        \
    acc = BX_MFMA(xm, yh, acc); acc = BX_MFMA(xh, ym, acc); acc = BX_MFMA(xh, yh, acc);

// column slice of a row-major plane as a 32x32x16 operand: rows r0 + 8 lh + 0..7 of column c0 + li.  TR: two ds_read_b64_tr_b16 (lane q
// of a 16-lane group passes the address of (row + q / 4, 4 (q % 4)) and receives column q of four rows); else eight 2-byte reads.
template <bool TR>
__device__ __forceinline__ bu32x4 plane_column8(const unsigned short* pl, int r0, int c0, int li, int lh) {
    bu32x4 v;
    if (TR) {
        const int i = li & 15;
        const unsigned short* at = pl + (r0 + 8 * lh + (i >> 2)) * BPL + c0 + 16 * (li >> 4) + 4 * (i & 3);
        typedef __attribute__((address_space(3))) bs16x4* lptr;
        const bs16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)at);
        const bs16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(at + 4 * BPL));
        const bu32x2 a = __builtin_bit_cast(bu32x2, t0), b = __builtin_bit_cast(bu32x2, t1);
        v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
    } else {
        const unsigned short* at = pl + (r0 + 8 * lh) * BPL + c0 + li;
        v.x = (unsigned)at[0] | ((unsigned)at[BPL] << 16);
        v.y = (unsigned)at[2 * BPL] | ((unsigned)at[3 * BPL] << 16);
        v.z = (unsigned)at[4 * BPL] | ((unsigned)at[5 * BPL] << 16);
        v.w = (unsigned)at[6 * BPL] | ((unsigned)at[7 * BPL] << 16);
    }
    return v;
}

// four consecutive columns of a row -> the three planes (8-byte stores)
__device__ __forceinline__ void plane_store4(unsigned short* pl, int o, float v0, float v1, float v2, float v3) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(v0, v1, h0, m0, l0);
    split3_pair(v2, v3, h1, m1, l1);
    *reinterpret_cast<uint2*>(pl + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(pl + BPLANE + o) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(pl + 2 * BPLANE + o) = make_uint2(l0, l1);
}

// K-split: the two waves that own the same 32 columns (one per 32-row block) split the k-range of the two weight-streamed products instead of the
// rows: each streams HALF of the tile's fragment planes and forms partial sums for BOTH row blocks, the halves meet through LDS (one
// extra barrier per product).  Without it both waves request the same 48 KB and the workgroup pulls 96 KB per direction through a CU
// that takes ~10 bytes per clock from L2: the forward product then ends when the stream does (11.9 k cycles into the launch).
// KSF / KSB: the forward / the backward-data product.  DEFAULT: backward only.  Forward, the other summation order is harmless as
// arithmetic (one more float32 addition per element) but h2 feeds a DISCONTINUOUS function, act'(h2): of the ~2 M pre-activations of a
// minibatch about 0.2 lie within float32 summation noise of zero, and an evaluation whose noise lands such an element on the other
// side than exact arithmetic moves one term of the 8 192-row gradient sums by 99 % -- 1e-4 of a cancelling tensor's scale.  The
// reference's own float32 gradient sits 6e-5 ... 1.5e-4 from its float64 twin on the C2 fixture for exactly that reason; the
// row-split forward happens to agree with the twin on every element of the fixtures (3e-7), the k-split one flips one
// (tools/probe_split_noise.py: critic.values.0.weight 2.5e-4) -- a coin any float32 evaluation tosses, but the fixtures are held
// where they are.  dH1 feeds only sums (g1 = dH1 * act'(h1), h1 untouched): the k-split there stays at 1e-7.
// LB: the barrier behind the small gradients moved behind the two matrix phases that follow them (a wave that is through with its share
// of the vector sums starts on the weight gradient at once).
// DS / AS: compile-time observation / head width (the CartPole class (4, 2): the headline) or 0 = from the arguments, D <= 8, A <= 4
// (Acrobot 6 / 3, LunarLander 8 / 4, MountainCar 2 / 3: every categorical classic-control / Box2D yaml of the reference).
// HEAD: 0 categorical, 1 Gaussian (mean as is), 2 Gaussian (tanh on the mean) -- ppo_trunk_kernel's heads (distributions.py:128-192).
// FWD: the forward pass only -- the ACTING pass of the on-policy loop for these networks (policies/categorical.py, gaussian.py
// ActorCriticPolicy.forward: representation -> actor head | critic head) as ONE launch instead of the three of the layered path: row m
// of f_obs [M][D] in, fwd_out[m][0..A) = the actor's output (activation_action applied), fwd_out[m][A] = the value.  Same products,
// same planes; nothing of the loss / backward phases is instantiated.
// With smp.n > 0 the forward-only launch also does what xrl_policy_sample does behind the layered forward (OnPolicyAgent.get_actions,
// core/on_policy.py:128-169; the same statements: csrc/sample.h): rows [0, n) are the observations -- the actor role samples the action
// and its log-prob, the critic role writes the value --, rows [n, 2 n) the previous step's next observations, whose value is bootv[t - 1]
// (their actor workgroups have nothing to do and leave).
template <bool FWD> struct bx_fwd_arg { typedef xrl_sample_t type; };
struct bx_no_sample { int unused; };
template <> struct bx_fwd_arg<false> { typedef bx_no_sample type; };

template <int ACT, bool TR, bool KSF, bool KSB, bool LB = false, int DS = 4, int AS = 2, int HEAD = 0, bool FWD = false>
__global__ void __launch_bounds__(FUSED_THREADS) ppo_trunk_bx_kernel(xrl_ppo_fused_t p, typename bx_fwd_arg<FWD>::type smp) {
    constexpr int DM = DS ? DS : BDMAX, AM = AS ? AS : BAMAX;
    constexpr bool GAUSS = HEAD != 0;
    constexpr int OACT = HEAD == 2 ? XRL_ACT_TANH : XRL_ACT_NONE;
    using L = BxLds<DM, AM>;
    constexpr int NQF = KSF ? 4 : 8, NQB = KSB ? 4 : 8, NQ = NQF > NQB ? NQF : NQB;     // k-steps of the weight-streamed products per wave
    constexpr int TPR = FUSED_THREADS / BPT;           // threads per row in the VALU phases: 8
    constexpr int NCH = (BH / 4) / TPR;                // float4 chunks of the branch level per thread: 4
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned short* h1p = reinterpret_cast<unsigned short*>(lds_raw + L::H1P);   // [3][64][136] h1 as bf16 planes
    unsigned short* g2p = reinterpret_cast<unsigned short*>(lds_raw + L::G2P);   // [3][64][136] dLoss/dz2 as bf16 planes
    float* h2 = reinterpret_cast<float*>(lds_raw + L::H2);                        // [64][132] (later: the row blocks' first-layer partial sums)
    float* xs = reinterpret_cast<float*>(lds_raw + L::XS);                        // [64][8] gathered observations, zero beyond D
    float* rsc = reinterpret_cast<float*>(lds_raw + L::RSC);                      // [64][12] act | . | ret | adv | old_logp
    float* dzh = reinterpret_cast<float*>(lds_raw + L::DZH);                      // [64][16] dLoss/d(head pre-activations)
    float* w0t = reinterpret_cast<float*>(lds_raw + L::W0T);                      // [4][132] first-layer weights, k-major
    float* b0s = reinterpret_cast<float*>(lds_raw + L::B0);
    float* bms = reinterpret_cast<float*>(lds_raw + L::BM);                       // this role's branch bias
    float* whs = reinterpret_cast<float*>(lds_raw + L::WH);                       // [nout][132] this role's head rows
    float* bhs = reinterpret_cast<float*>(lds_raw + L::BHS);
    float* lss = reinterpret_cast<float*>(lds_raw + L::LS);                       // log_std
    double* rowstat = reinterpret_cast<double*>(lds_raw + L::RST);                // [5][64] per-row loss terms
    double* bgp = reinterpret_cast<double*>(lds_raw + L::BGP);                    // [2][128] branch-bias gradient of the two row halves

    kernarg_prefetch<sizeof(xrl_ppo_fused_t) + (FWD ? sizeof(xrl_sample_t) : 0)>();
    const int tid = threadIdx.x, M = p.M, D = DS ? DS : p.D, A = AS ? AS : p.A;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cblk = wave & 3, rblk = wave >> 2;       // matrix phases: 32-column block / 32-row block of this wave
    // (role = parity of the workgroup index = parity of its XCD: an XCD's L2 then holds ONE role's fragment planes after the optimiser launch's
    //  write-back.  Measured: the actor role -- the longer one -- on the first half of the grid, for the head start of the dispatch order,
    //  puts both roles' planes through every L2: 20.9 -> 19.9 M env-steps/s.)
    const int tile = blockIdx.x >> 1, role = blockIdx.x & 1;
    const bool actor = role == 0;
    const int nout = actor ? A : 1;
    const int cb = role * BH;                          // this role's first column of the stacked branch level
    const int m0 = tile * BPT;
    if constexpr (FWD) {
        if (smp.n > 0 && actor && (m0 >= smp.n || !smp.act_out)) return;   // (uniform: no action is sampled for these rows)
    }
    const int r = tid / TPR, sub = tid % TPR, m_row = m0 + r;
    const bool row_ok = m_row < M;
    float* slab = p.slabs + (size_t)tile * p.slab_stride;
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];
    const xrl_fused_layer_t& Lh = actor ? La : Lc;

    long long* dbg = p.dbg;                            // diagnostics (tools/probe_pair_phases.py)
    const bool dbg_me = dbg && tid == 0 && blockIdx.x == gridDim.x - (p.pad3 & 2 ? 2 : 1);   // (pad3 bit 1: stamp an ACTOR workgroup)
#define TSTAMP(k) do { if (dbg_me) dbg[k] = clock64(); } while (0)
    if (dbg && tid == 0) dbg[16 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    TSTAMP(0);

    // ================= loads: EVERY request first (record rows, the three small-parameter words of this thread, this role's forward
    // fragment planes: 8 k-steps x 3 planes x 16 bytes per lane), then the LDS writes -- written as loops over the parameter arrays the
    // compiler emits load -> s_waitcnt vmcnt(0) -> ds_write per array: five L2 round trips in a row in front of the fragment stream
    const bool records = !FWD && !GAUSS && D == 4 && (p.f_rows || p.f_packed);
    float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (records) {                                     // 32-byte records obs[4] | act | ret | adv | old_logp: one wave
        if (wave == 7) {
            const int m = m0 + lane;
            if (m < M) {
                size_t at = (size_t)m;
                if (!p.f_rows) {
                    const int64_t fl = p.idx[m];
                    const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
                    at = (size_t)t * p.n_envs + env;
                }
                const float4* rec = reinterpret_cast<const float4*>(p.f_rows ? p.f_rows : p.f_packed) + at * 2;
                xr = rec[0]; sc = rec[1];
            }
        }
    }
    // rows through idx from the buffer's FIELDS: thread (row gr = tid / 8, gs = tid % 8) takes the row's observation element gs and one
    // of its scalars (gs < na: action component, 4 return, 5 advantage, 6 old log-prob) -- every thread resolves its row's buffer index
    // itself and all gathers of the tile are in flight together (as loops over the tile behind a shared index array they were three
    // dependent round trips to rows scattered over the buffer: 28.7 us per launch for Acrobot where the record path takes 22.1)
    float g_x = 0.f, g_s = 0.f;
    const int gr = tid >> 3, gs = tid & 7;
    if (!records) {
        const int m = m0 + gr;
        int src = -1;
        if (m < M) {
            if (FWD) src = m;
            else {                                     // env-major flat index (memory_tools.py:270) -> time-major buffer row
                const int64_t fl = p.idx[m];
                const int env = (int)(fl / p.T), t = (int)(fl - (int64_t)env * p.T);
                src = t * p.n_envs + env;
            }
        }
        if (src >= 0) {
            if (gs < D) g_x = p.f_obs[(size_t)src * D + gs];
            if (!FWD) {
                const int na = GAUSS ? A : 1;
                if (gs < na) g_s = p.f_act[(size_t)src * na + gs];
                else if (gs == 4) g_s = p.f_ret[src];
                else if (gs == 5) g_s = p.f_adv[src];
                else if (gs == 6) g_s = p.f_logp[src];
            }
        }
    }
    float st_mean = 0.f, st_std = 1.f;
    if (!FWD && p.stats) { st_mean = p.stats[0]; st_std = p.stats[1]; }
    // W0 [128][D]: one (D = 4) or two elements per thread; b0 | this role's branch bias | head bias: one word of threads 0..263; head rows:
    // threads 0..nout*128
    constexpr int W0Q = (BH * DM + FUSED_THREADS - 1) / FUSED_THREADS;
    float w0v[W0Q];
#pragma unroll
    for (int q = 0; q < W0Q; ++q) {
        const int e = tid + q * FUSED_THREADS;
        w0v[q] = e < BH * D ? p.params[L0.w_off + e] : 0.f;
    }
    float smv = 0.f;
    if (tid < BH) smv = p.params[L0.b_off + tid];
    else if (tid < 2 * BH) smv = p.params[L1.b_off + cb + tid - BH];
    else if (tid < 2 * BH + nout) smv = p.params[Lh.b_off + tid - 2 * BH];
    float lsv = 0.f;
    if (GAUSS && tid >= 2 * BH && tid < 2 * BH + A) lsv = p.params[p.log_std_off + tid - 2 * BH];
    float whv = 0.f;
    if (tid < nout * BH) whv = p.params[Lh.w_off + tid];
    bu32x4 pf[NQ][3];
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.frag16), 0, 3 * XRL_FRAG16_PLANE * 2, 0x00020000);
    {
        const int t = 4 * role + cblk;                                   // 32-row tile of the stacked 256-row W1
        // (the first half of the stream only: a wave cannot reach its LDS writes -- and the workgroup its first barrier -- before all of
        //  its requests are ISSUED, and 24 x 1 KB per wave through one CU's address path is ~1.4 k cycles; the second half goes out
        //  behind the barrier, in front of the first layer, and is first needed after four k-steps of the branch layer)
#pragma unroll
        for (int j = 0; j < NQF / 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int qq = KSF ? 4 * rblk + j : j;
                pf[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (pl * XRL_FRAG16_PLANE + (t * 8 + ((qq + t) & 7)) * 512) * 2, 0);
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (records && wave == 7) {
        *reinterpret_cast<float4*>(xs + lane * BXLD) = xr;
        *reinterpret_cast<float4*>(xs + lane * BXLD + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        rsc[lane * 12 + 0] = sc.x; rsc[lane * 12 + 8] = sc.y; rsc[lane * 12 + 9] = sc.z; rsc[lane * 12 + 10] = sc.w;
    }
#pragma unroll
    for (int q = 0; q < W0Q; ++q) {                                      // element e = column e / D, k = e % D -> k-major
        const int e = tid + q * FUSED_THREADS;
        if (e < BH * D) { const int c = e / D, k = e - c * D; w0t[k * BLD + c] = w0v[q]; }
    }
    if (!DS && (D & 1) && tid < BH) w0t[D * BLD + tid] = 0.f;           // (the first layer's MFMAs walk k in pairs)
    if (tid < BH) b0s[tid] = smv;
    else if (tid < 2 * BH) bms[tid - BH] = smv;
    else if (tid < 2 * BH + 8) { bhs[tid - 2 * BH] = smv; lss[tid - 2 * BH] = lsv; }
    if (tid < nout * BH) whs[(tid >> 7) * BLD + (tid & (BH - 1))] = whv;
    if (!records) {
        xs[gr * BXLD + gs] = g_x;                                        // (zero beyond D and for rows past M)
        if (!FWD) {
            if (gs < 4) rsc[gr * 12 + gs] = g_s;                         // action components (zero beyond na)
            else if (gs < 7) rsc[gr * 12 + 4 + gs] = g_s;                // 8 return | 9 advantage | 10 old log-prob
        }
    }
    lds_barrier();                                                                                   // #0 rows, parameters
    TSTAMP(1);
    {
        const int t = 4 * role + cblk;
#pragma unroll
        for (int j = NQF / 2; j < NQF; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int qq = KSF ? 4 * rblk + j : j;
                pf[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (pl * XRL_FRAG16_PLANE + (t * 8 + ((qq + t) & 7)) * 512) * 2, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ================= forward.  First layer, transposed: C[i = column n][j = row] = sum_k W0[n][k] x[row][k] -- ppo_trunk_kernel's two
    // fp32 instructions with the operands exchanged (the same products in the same order); lane (li = row, lh) then holds columns
    // 32 cblk + 8 g + 4 lh + 0..3 (g = 0..3): four consecutive k of the next product -> split once, three 8-byte stores per group
    {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* wk = w0t + lh * BLD + cblk * 32 + li;              // A[i = column][k = lh + 2 s]
        const float* xr = xs + (rblk * 32 + li) * BXLD + lh;            // B[k = lh + 2 s][j = row]
        if (DS == 4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[0], xr[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[2 * BLD], xr[2], acc, 0, 0, 0);
        } else {
            const int ns = (D + 1) >> 1;
            for (int s2 = 0; s2 < ns; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[2 * s2 * BLD], xr[2 * s2], acc, 0, 0, 0);
        }
        const int row = rblk * 32 + li;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = cblk * 32 + 8 * g + 4 * lh;
            const float4 b = *reinterpret_cast<const float4*>(b0s + n0);
            plane_store4(h1p, row * BPL + n0, act_apply_c<ACT>(acc[4 * g] + b.x), act_apply_c<ACT>(acc[4 * g + 1] + b.y),
                         act_apply_c<ACT>(acc[4 * g + 2] + b.z), act_apply_c<ACT>(acc[4 * g + 3] + b.w));
        }
    }
    lds_barrier();                                                                                   // #1 h1
    TSTAMP(2);
    // ---- this role's branch layer 128 -> 128, transposed as well: wave (cblk, rblk): C[i = column 32 cblk + ..][j = row 32 rblk + li];
    //      A = the weight planes in registers, B = h1's planes (16-byte row reads: k = 16 qq + 8 lh + 0..7)
    {
        const int row = rblk * 32 + li;                                 // the row this lane finishes
        if constexpr (!KSF) {
            const unsigned short* hrow = h1p + row * BPL + 8 * lh;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const bu32x4 xh = *reinterpret_cast<const bu32x4*>(hrow + 16 * qq), xm = *reinterpret_cast<const bu32x4*>(hrow + BPLANE + 16 * qq),
                             xl = *reinterpret_cast<const bu32x4*>(hrow + 2 * BPLANE + 16 * qq);
                // (term names: activation part, weight part -- the operands of the instruction are (weights, rows))
                acc = BX_MFMA(pf[qq][0], xl, acc); acc = BX_MFMA(pf[qq][2], xh, acc); acc = BX_MFMA(pf[qq][1], xm, acc);
                acc = BX_MFMA(pf[qq][0], xm, acc); acc = BX_MFMA(pf[qq][1], xh, acc); acc = BX_MFMA(pf[qq][0], xh, acc);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = cblk * 32 + 8 * g + 4 * lh;
                const float4 b = *reinterpret_cast<const float4*>(bms + n0);
                *reinterpret_cast<float4*>(h2 + row * BLD + n0) = make_float4(act_apply_c<ACT>(acc[4 * g] + b.x), act_apply_c<ACT>(acc[4 * g + 1] + b.y),
                                                                             act_apply_c<ACT>(acc[4 * g + 2] + b.z), act_apply_c<ACT>(acc[4 * g + 3] + b.w));
            }
        } else {
            // k-steps 4 rblk .. 4 rblk + 3 for both row blocks: `mine` = this wave's block (rows 32 rblk + li), `other` = the partner's
            const int orow = (1 - rblk) * 32 + li;
            const unsigned short* hmine = h1p + row * BPL + 8 * lh + 64 * rblk;
            const unsigned short* hother = h1p + orow * BPL + 8 * lh + 64 * rblk;
            f32x16 mine, other;
#pragma unroll
            for (int i = 0; i < 16; ++i) { mine[i] = 0.f; other[i] = 0.f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bu32x4 xh = *reinterpret_cast<const bu32x4*>(hmine + 16 * j), xm = *reinterpret_cast<const bu32x4*>(hmine + BPLANE + 16 * j),
                             xl = *reinterpret_cast<const bu32x4*>(hmine + 2 * BPLANE + 16 * j);
                const bu32x4 yh = *reinterpret_cast<const bu32x4*>(hother + 16 * j), ym = *reinterpret_cast<const bu32x4*>(hother + BPLANE + 16 * j),
                             yl = *reinterpret_cast<const bu32x4*>(hother + 2 * BPLANE + 16 * j);
                mine = BX_MFMA(pf[j][0], xl, mine); other = BX_MFMA(pf[j][0], yl, other);
                mine = BX_MFMA(pf[j][2], xh, mine); other = BX_MFMA(pf[j][2], yh, other);
                mine = BX_MFMA(pf[j][1], xm, mine); other = BX_MFMA(pf[j][1], ym, other);
                mine = BX_MFMA(pf[j][0], xm, mine); other = BX_MFMA(pf[j][0], ym, other);
                mine = BX_MFMA(pf[j][1], xh, mine); other = BX_MFMA(pf[j][1], yh, other);
                mine = BX_MFMA(pf[j][0], xh, mine); other = BX_MFMA(pf[j][0], yh, other);
            }
            // the partner's half of `other` rows goes where their h2 will be; it leaves its half of this wave's rows there
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(h2 + orow * BLD + cblk * 32 + 8 * g + 4 * lh) = make_float4(other[4 * g], other[4 * g + 1], other[4 * g + 2], other[4 * g + 3]);
            lds_barrier();                                                                           // #1b the k-halves
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = cblk * 32 + 8 * g + 4 * lh;
                const float4 b = *reinterpret_cast<const float4*>(bms + n0), o = *reinterpret_cast<const float4*>(h2 + row * BLD + n0);
                // (k-half 0 + k-half 1, whichever wave adds them)
                *reinterpret_cast<float4*>(h2 + row * BLD + n0) = make_float4(act_apply_c<ACT>((mine[4 * g] + o.x) + b.x), act_apply_c<ACT>((mine[4 * g + 1] + o.y) + b.y),
                                                                             act_apply_c<ACT>((mine[4 * g + 2] + o.z) + b.z), act_apply_c<ACT>((mine[4 * g + 3] + o.w) + b.w));
            }
        }
        // forward planes consumed: the same registers take the BACKWARD section (output tile kt = cblk of dH1, this role's 8 n-steps
        // q = 8 role + qq), which has the head / loss / weight-gradient phases to arrive
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < (FWD ? 0 : NQB); ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int q = 8 * role + (KSB ? 4 * rblk + j : j);
                pf[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, (pl * XRL_FRAG16_PLANE + XRL_FRAG16_PLANE / 2 + (cblk * 16 + ((q + cblk) & 15)) * 512) * 2, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();                                                                                   // #2 h2
    TSTAMP(3);

    // ================= head forward (VALU, 8 threads per row), this role's loss terms, head backward -- in registers
    float4 a[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) a[i] = *reinterpret_cast<const float4*>(h2 + r * BLD + 4 * (sub + TPR * i));
    float z[AM];
#pragma unroll
    for (int j = 0; j < AM; ++j) {
        z[j] = 0.f;
        if (j < nout) {
            float c = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(whs + j * BLD + 4 * (sub + TPR * i));
                c += a[i].x * w.x + a[i].y * w.y + a[i].z * w.z + a[i].w * w.w;
            }
            c += __shfl_xor(c, 4, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 1, 64);
            z[j] = c + bhs[j];
        }
    }
    if constexpr (FWD) {
        if (sub == 0 && row_ok) {
            if (smp.n > 0) {                                             // heads + xrl_policy_sample's work, no head buffer
                if (actor) {
                    if (m_row < smp.n && smp.act_out) {
                        float h[AM];
#pragma unroll
                        for (int j = 0; j < AM; ++j) h[j] = act_apply_c<OACT>(z[j]);
                        policy_sample_actor(smp, m_row, h);
                    }
                } else if (m_row < smp.n) {
                    if (smp.act_out && smp.val_out) smp.val_out[m_row] = z[0];
                } else if (smp.bootv_prev) smp.bootv_prev[m_row - smp.n] = z[0];
            } else {
                float* out = p.fwd_out + (size_t)m_row * p.fwd_ld;
                if (actor) {
#pragma unroll
                    for (int j = 0; j < AM; ++j) if (j < A) out[j] = act_apply_c<OACT>(z[j]);
                } else out[A] = z[0];
            }
        }
        return;
    }
    {
        float dz[AM];
#pragma unroll
        for (int j = 0; j < AM; ++j) dz[j] = 0.f;
        double t_s = 0.0, t_c = 0.0, t_e = 0.0, t_v = 0.0, t_n = 0.0;
        const float invM = 1.f / (float)M;
        if (actor) {
            float adv = rsc[r * 12 + 9];
            const float old_lp = rsc[r * 12 + 10];
            asm volatile("" : "+v"(st_std));
            if (p.stats) adv = __fdiv_rn(__fsub_rn(adv, st_mean), st_std + 1e-8f);                   // memory_tools.py:281-282
            const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
            if constexpr (!GAUSS) {
            if (row_ok) {
                const int act = (int)rsc[r * 12];
                float mx = z[0];
#pragma unroll
                for (int j = 1; j < AM; ++j) if (j < A) mx = fmaxf(mx, z[j]);
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < AM; ++j) if (j < A) se += expf(z[j] - mx);
                const float lse = mx + logf(se);
                float zact = z[0];
#pragma unroll
                for (int j = 1; j < AM; ++j) if (j == act) zact = z[j];
                const float logp = zact - lse;
                float ent = 0.f;
#pragma unroll
                for (int j = 0; j < AM; ++j) if (j < A) { const float l = z[j] - lse; ent -= expf(l) * l; }
                const Surrogate s = surrogate(logp, old_lp, adv, lo, hi, invM);
                const float ce = p.ent_coef * invM;
#pragma unroll
                for (int j = 0; j < AM; ++j)
                    if (j < A) { const float l = z[j] - lse, pj = expf(l); dz[j] = s.dlogp * ((j == act ? 1.f : 0.f) - pj) + ce * pj * (l + ent); }
                t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                if (p.diag && sub == 0) {
                    const int m = m_row;
                    p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                }
            }
            } else {
                // DiagGaussianDistribution (distributions.py:155-192): log_prob / entropy summed over the action dims; thread `sub` of a row
                // owns dimension `sub` (ppo_trunk_kernel's statements)
                const bool mine = sub < A;
                float zmine = 0.f;
#pragma unroll
                for (int j = 0; j < AM; ++j) if (j == sub) zmine = z[j];
                float df = 0.f, var = 1.f, term = 0.f, entj = 0.f, mu = 0.f;
                if (mine) {
                    mu = act_apply_c<OACT>(zmine);                                                   // activation_action (actor_head.py:62)
                    const float ls = lss[sub], sd = expf(ls);
                    var = sd * sd; df = rsc[r * 12 + sub] - mu;
                    term = -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;
                    entj = 0.5f + LOG_SQRT_2PI + logf(sd);
                }
                float logp = term, ent = entj;
                logp += __shfl_xor(logp, 4, 64); logp += __shfl_xor(logp, 2, 64); logp += __shfl_xor(logp, 1, 64);
                ent += __shfl_xor(ent, 4, 64); ent += __shfl_xor(ent, 2, 64); ent += __shfl_xor(ent, 1, 64);
                float my_dz = 0.f, my_gls = 0.f;
                if (row_ok) {
                    const Surrogate s = surrogate(logp, old_lp, adv, lo, hi, invM);
                    if (mine) {
                        my_dz = (s.dlogp * df / var) * act_grad_c<OACT>(mu);
                        my_gls = s.dlogp * (df * df / var - 1.f);
                    }
                    t_s = (double)fminf(s.s1, s.s2); t_n = s.clipped; t_e = ent;
                    if (p.diag && sub == 0) {
                        const int m = m_row;
                        p.diag[m] = logp; p.diag[M + m] = s.ratio; p.diag[2 * (size_t)M + m] = s.s1; p.diag[3 * (size_t)M + m] = s.s2;
                    }
                }
                dzh[r * 16 + sub] = my_dz; dzh[r * 16 + 8 + sub] = my_gls;          // (8 threads per row = the 8 + 8 slots)
                // the row's threads sit in one wave and a wave's LDS operations execute in order: the values are there
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float4 d0 = *reinterpret_cast<const float4*>(dzh + r * 16);
                dz[0] = d0.x;
                if (AM > 1) dz[1 < AM ? 1 : 0] = d0.y;
                if (AM > 2) dz[2 < AM ? 2 : 0] = d0.z;
                if (AM > 3) dz[3 < AM ? 3 : 0] = d0.w;
            }
        } else if (row_ok) {
            const float v = z[0], dv = v - rsc[r * 12 + 8];
            dz[0] = p.vf_coef * 2.f * dv * invM;
            t_c = (double)dv * dv; t_v = v;
        }
        if (sub == 0) {
            if (!(actor && GAUSS)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { dzh[r * 16 + j] = j < AM ? dz[j < AM ? j : 0] : 0.f; dzh[r * 16 + 8 + j] = 0.f; }
            }
            rowstat[0 * BPT + r] = t_s; rowstat[1 * BPT + r] = t_c; rowstat[2 * BPT + r] = t_e; rowstat[3 * BPT + r] = t_v; rowstat[4 * BPT + r] = t_n;
        }
        // g2 = (dZh . W_h) * act'(h2): this thread's k-chunks, split where they are made
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < AM; ++j) {
                if (j < nout) {
                    const float4 w = *reinterpret_cast<const float4*>(whs + j * BLD + 4 * (sub + TPR * i));
                    g.x += dz[j] * w.x; g.y += dz[j] * w.y; g.z += dz[j] * w.z; g.w += dz[j] * w.w;
                }
            }
            g.x *= act_grad_c<ACT>(a[i].x); g.y *= act_grad_c<ACT>(a[i].y); g.z *= act_grad_c<ACT>(a[i].z); g.w *= act_grad_c<ACT>(a[i].w);
            plane_store4(g2p, r * BPL + 4 * (sub + TPR * i), g.x, g.y, g.z, g.w);
        }
    }
    lds_barrier();                                                                                   // #3 g2, dzh, rowstat
    TSTAMP(4);

    // ================= backward
    // ---- loss terms of this (tile, role): one statistic per wave (see ppo_trunk_kernel)
    if (wave < 5) {
        double s = rowstat[wave * BPT + lane];
        s = wave_sum(s);
        if (lane == 0) p.partials[(size_t)blockIdx.x * 8 + wave] = s;
    } else if (wave == 5 && lane < 3) p.partials[(size_t)blockIdx.x * 8 + 5 + lane] = 0.0;
    // ---- head weight / bias gradients, this role's branch-layer bias gradient: VALU sums over the 64 rows, rows in order
    for (int e = tid; e < nout * BH; e += FUSED_THREADS) {
        const int j = e >> 7, k = e & (BH - 1);
        const float* hp = h2 + k;
        float acc = 0.f;
#pragma unroll 16
        for (int rr = 0; rr < BPT; ++rr) acc += dzh[rr * 16 + j] * hp[rr * BLD];
        slab[Lh.w_off + e] = acc;
    }
    // (column sums in double, rounded once: ppo_trunk_kernel's note.  The branch bias: waves 4..7, column t, rows [32 rh, +32) -- each
    //  element comes back from its three planes exactly; the two halves meet behind the barrier.  The head rows' loops above run on
    //  waves 0..3 / 0..1 meanwhile.)
    if (tid >= 4 * 64) {
        const int t = (tid - 4 * 64) & (BH - 1), rh = (tid - 4 * 64) >> 7;
        double acc0 = 0.0;
#pragma unroll 16
        for (int rr = 0; rr < BPT / 2; ++rr) acc0 += (double)plane_value(g2p, (rh * (BPT / 2) + rr) * BPL + t);
        bgp[rh * BH + t] = acc0;
    } else if (tid >= 3 * 64 && tid < 3 * 64 + 16) {
        const int t = tid - 3 * 64;                                      // 0..7 head bias, 8..15 log_std
        if (t < nout || (t >= 8 && GAUSS && actor && t - 8 < A)) {
            double acc = 0.0;
#pragma unroll 16
            for (int rr = 0; rr < BPT; ++rr) acc += (double)dzh[rr * 16 + t];
            // d(-ent_coef * mean_m sum_j(log_std_j + c)) / d log_std_j = -ent_coef, added once (tile 0), as in ppo_trunk_kernel
            if (t >= 8 && tile == 0) acc -= (double)p.ent_coef;
            if (t < 8) slab[Lh.b_off + t] = (float)acc;
            else slab[p.log_std_off + t - 8] = (float)acc;
        }
    }
    if constexpr (!LB) {
        lds_barrier();                                                                               // #3b no matrix instruction beside a vector loop
        if (tid < BH) slab[L1.b_off + cb + tid] = (float)(bgp[tid] + bgp[BH + tid]);
    }
    TSTAMP(5);
    // ---- dW1[n][k] = sum over the 64 rows of g2[row][n] * h1[row][k] for this role's 128 rows n: 4 x 4 tiles of 32 x 32, wave w owns
    //      n-tile (w & 3) and the k-tiles 2 (w >> 2), 2 (w >> 2) + 1; four 16-row steps; both operands are column slices of planes
    //      (the instruction's A operand is h1's slice, B is g2's: the tile comes out as [k][n])
    f32x16 dacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) dacc[i] = 0.f;
    {
        const int nt = wave & 3, kt0 = 2 * (wave >> 2);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int rs = 0; rs < 4; ++rs) {
            const bu32x4 gh = plane_column8<TR>(g2p, 16 * rs, nt * 32, li, lh), gm = plane_column8<TR>(g2p + BPLANE, 16 * rs, nt * 32, li, lh),
                         gl = plane_column8<TR>(g2p + 2 * BPLANE, 16 * rs, nt * 32, li, lh);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int c0 = (kt0 + t) * 32;
                const bu32x4 xh = plane_column8<TR>(h1p, 16 * rs, c0, li, lh), xm = plane_column8<TR>(h1p + BPLANE, 16 * rs, c0, li, lh),
                             xl = plane_column8<TR>(h1p + 2 * BPLANE, 16 * rs, c0, li, lh);
                BX_MFMA6(xh, xm, xl, gh, gm, gl, acc[t])                // (transposed: C[i = k][j = n] -- see the stores below)
            }
        }
        TSTAMP(6);
        // ---- this role's part of dH1 = g2 . W1 (sum over its 128 rows n): wave (cblk, rblk) owns output columns [32 cblk, +32) of
        //      rows [32 rblk, +32); A = g2's planes (row reads), B = the backward planes requested after the forward layer.
        //      The 32 stores of this wave's dW1 tiles are issued between these instructions (ppo_trunk_kernel's note).
        float* dW = slab + L1.w_off + (size_t)(cb + nt * 32) * BH;
        f32x16 dother;
        if constexpr (!KSB) {
            const unsigned short* grow = g2p + (rblk * 32 + li) * BPL + 8 * lh;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const bu32x4 gh = *reinterpret_cast<const bu32x4*>(grow + 16 * qq), gm = *reinterpret_cast<const bu32x4*>(grow + BPLANE + 16 * qq),
                             gl = *reinterpret_cast<const bu32x4*>(grow + 2 * BPLANE + 16 * qq);
                BX_MFMA6(gh, gm, gl, pf[qq][0], pf[qq][1], pf[qq][2], dacc)
            }
        } else {
            // n-steps 4 rblk .. 4 rblk + 3 of this role's 8, both row blocks (dacc: this wave's rows, dother: the partner's)
#pragma unroll
            for (int i = 0; i < 16; ++i) dother[i] = 0.f;
            const unsigned short* gmine = g2p + (rblk * 32 + li) * BPL + 8 * lh + 64 * rblk;
            const unsigned short* goth = g2p + ((1 - rblk) * 32 + li) * BPL + 8 * lh + 64 * rblk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bu32x4 gh = *reinterpret_cast<const bu32x4*>(gmine + 16 * j), gm = *reinterpret_cast<const bu32x4*>(gmine + BPLANE + 16 * j),
                             gl = *reinterpret_cast<const bu32x4*>(gmine + 2 * BPLANE + 16 * j);
                const bu32x4 oh = *reinterpret_cast<const bu32x4*>(goth + 16 * j), om = *reinterpret_cast<const bu32x4*>(goth + BPLANE + 16 * j),
                             ol = *reinterpret_cast<const bu32x4*>(goth + 2 * BPLANE + 16 * j);
                dacc = BX_MFMA(gl, pf[j][0], dacc); dother = BX_MFMA(ol, pf[j][0], dother);
                dacc = BX_MFMA(gh, pf[j][2], dacc); dother = BX_MFMA(oh, pf[j][2], dother);
                dacc = BX_MFMA(gm, pf[j][1], dacc); dother = BX_MFMA(om, pf[j][1], dother);
                dacc = BX_MFMA(gm, pf[j][0], dacc); dother = BX_MFMA(om, pf[j][0], dother);
                dacc = BX_MFMA(gh, pf[j][1], dacc); dother = BX_MFMA(oh, pf[j][1], dother);
                dacc = BX_MFMA(gh, pf[j][0], dacc); dother = BX_MFMA(oh, pf[j][0], dother);
            }
        }
        // (the weight-gradient tiles were formed TRANSPOSED -- A = h1's column slices, B = g2's -- so that lane (li, lh) holds
        //  dW1[n = 32 nt + li][k = 32 kt + 8 g + 4 lh + 0..3]: four consecutive floats of a slab row, 8 stores of 16 bytes per lane
        //  instead of 32 of 4.  The slab write-out is bound by store ISSUE on this chip (MI355X_MICROARCH.md: a row-per-lane epilogue
        //  of narrow stores drains at ~7 bytes per clock and CU), and 64 KB per workgroup at that rate was the longest part of this phase.)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(dW + (size_t)li * BH + (kt0 + t) * 32 + 8 * g + 4 * lh) =
                    make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
#pragma unroll
        for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); }
        TSTAMP(9);
        if constexpr (LB) {
            lds_barrier();                                                                           // #3b, late: h2 / dzh / rsc have no reader left
            TSTAMP(10);
            if (tid < BH) slab[L1.b_off + cb + tid] = (float)(bgp[tid] + bgp[BH + tid]);
        }
        if constexpr (KSB) {
            // the n-halves meet where h2 was (its last readers are behind barrier #3b): [wave][4][64 lanes] float4
            float4* xch = reinterpret_cast<float4*>(h2);
#pragma unroll
            for (int g = 0; g < 4; ++g) xch[(wave * 4 + g) * 64 + lane] = make_float4(dother[4 * g], dother[4 * g + 1], dother[4 * g + 2], dother[4 * g + 3]);
            lds_barrier();                                                                           // #3c the n-halves
            TSTAMP(11);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = xch[((wave ^ 4) * 4 + g) * 64 + lane];
                dacc[4 * g] += o.x; dacc[4 * g + 1] += o.y; dacc[4 * g + 2] += o.z; dacc[4 * g + 3] += o.w;
            }
        }
    }
    // ---- first layer: dW0[c][k] = sum_rows g1[row][c] * x[row][k], db0[c], g1 = dH1 * act'(h1) -- straight from the dH1 accumulators
    //      (ppo_trunk_kernel's form; h1 comes back from its planes, exactly)
    {
        float* dst = actor ? slab : slab + p.l0_fold_off;
        const int w_at = actor ? L0.w_off : 0, b_at = actor ? L0.b_off : BH * D;
        // [2][128][D + 1] partial sums of the row blocks: where h2 was, or (KSB: h2's place holds the n-halves other waves may still be
        // reading) where the row scalars and head gradients were -- dead since barrier #3b
        float* part = KSB ? rsc : h2;
        // (BxLds: RSC, DZH, W0T are contiguous and hold [2][128][DM + 1] floats; the first-layer weights' last readers are behind barrier #1)
        constexpr int PLD = DM + 1;
        {
            const int c = cblk * 32 + li;
            float ab = 0.f;
            float acc[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) acc[k] = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int row = rblk * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                // (relu / leaky_relu: act' depends on the SIGN of h1 only, and h = bf16(h1) has h1's sign and is zero only where h1 is:
                //  one 2-byte read instead of three and two additions; tanh needs the value: all three planes, exactly)
                const float h1v = (ACT == XRL_ACT_RELU || ACT == XRL_ACT_LEAKY_RELU) ? bf16_bits_to_float(h1p[row * BPL + c])
                                                                                      : plane_value(h1p, row * BPL + c);
                const float g = dacc[rr] * act_grad_c<ACT>(h1v);
                ab += g;
#pragma unroll
                for (int q = 0; q < DM / 4; ++q) {
                    if (4 * q < D) {
                        const float4 x = *reinterpret_cast<const float4*>(xs + row * BXLD + 4 * q);      // (zero beyond D)
                        acc[4 * q] += g * x.x; acc[4 * q + 1] += g * x.y; acc[4 * q + 2] += g * x.z; acc[4 * q + 3] += g * x.w;
                    }
                }
            }
            // (h2's last readers -- the head weight gradient -- are behind barrier #3b)
#pragma unroll
            for (int k = 0; k < DM; ++k) acc[k] += __shfl_xor(acc[k], 32, 64);
            ab += __shfl_xor(ab, 32, 64);
            if (lh == 0) {
                float* pp = part + (rblk * BH + c) * PLD;
#pragma unroll
                for (int k = 0; k < DM; ++k) pp[k] = acc[k];
                pp[DM] = ab;
            }
        }
        TSTAMP(12);
        lds_barrier();                                                                               // #4 the two row blocks' partial sums
        TSTAMP(7);
        for (int e = tid; e < BH * (D + 1); e += FUSED_THREADS) {
            const int c = e / (D + 1), k = e - c * (D + 1), kk = k < D ? k : DM;
            const float v = part[c * PLD + kk] + part[(BH + c) * PLD + kk];
            dst[k < D ? w_at + c * D + k : b_at + c] = v;
        }
    }
    TSTAMP(8);
    if (dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[17 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#undef TSTAMP
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static int g_bx_tr = 1;                                   // weight-gradient operands through ds_read_b64_tr_b16 (0: 2-byte reads; diagnostics)
static int g_bx_ks = 5;                                   // the wave pair of a column block splits k, not rows: 0 nowhere, 1 backward-data product, 2 both

bool ppo_trunk_bx_eligible(const xrl_ppo_fused_t& p) {
    return p.frag16 != nullptr && p.pad0 == 64 && (p.dist == 0 || p.dist == 1) && p.D >= 1 && p.D <= BDMAX && p.A >= 1 && p.A <= BAMAX &&
           p.layers[1].N == 2 * BH && p.layers[1].K == BH;
}

template <int ACT>
static int launch_bx(const xrl_ppo_fused_t& p, hipStream_t stream) {
    const int n_tiles = (p.M + BPT - 1) / BPT;
    constexpr int LDSB = BxLds<4, 2>::BYTES, LDSG = BxLds<BDMAX, BAMAX>::BYTES;
    if (p.dist == 1) {                                    // Gaussian heads (Pendulum (3, 1)): any (D <= 8, A <= 4), the default form only
        if (p.out_act == XRL_ACT_TANH) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, true, true, 0, 0, 2>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSG, stream, p, bx_no_sample{0});
        else hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, true, true, 0, 0, 1>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSG, stream, p, bx_no_sample{0});
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    if (!(p.D == 4 && p.A == 2)) {                        // any (D <= 8, A <= 4): the default form only
        hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, true, true, 0, 0>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSG, stream, p, bx_no_sample{0});
        XRL_CHECK_LAUNCH();
        return XRL_OK;
    }
    if (!g_bx_tr) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, false, false, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSB, stream, p, bx_no_sample{0});
    else if (g_bx_ks == 1) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSB, stream, p, bx_no_sample{0});
    else if (g_bx_ks == 5) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, true, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSB, stream, p, bx_no_sample{0});
    else if (g_bx_ks == 2) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, true, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSB, stream, p, bx_no_sample{0});
    else hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, false>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSB, stream, p, bx_no_sample{0});
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

int launch_ppo_trunk_bx(const xrl_ppo_fused_t& p, hipStream_t stream) {
    switch (p.layers[0].act) {
        case XRL_ACT_RELU: return launch_bx<XRL_ACT_RELU>(p, stream);
        case XRL_ACT_LEAKY_RELU: return launch_bx<XRL_ACT_LEAKY_RELU>(p, stream);
        default: return launch_bx<XRL_ACT_TANH>(p, stream);
    }
}

template <int ACT>
static int init_bx_one() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<4, 2>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<4, 2>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<4, 2>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<4, 2>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<4, 2>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, true, true, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<BDMAX, BAMAX>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, true, true, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<BDMAX, BAMAX>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, true, true, 0, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<BDMAX, BAMAX>::BYTES));
    return XRL_OK;
}

int init_ppo_trunk_bx_fwd();
int init_ppo_trunk_bx() {
    if (int rc = init_ppo_trunk_bx_fwd()) return rc;
    if (int rc = init_bx_one<XRL_ACT_RELU>()) return rc;
    if (int rc = init_bx_one<XRL_ACT_LEAKY_RELU>()) return rc;
    if (int rc = init_bx_one<XRL_ACT_TANH>()) return rc;
    return XRL_OK;
}

// image <- W[256][128] of the stacked branch layer as three bf16 planes in the order the kernel's lanes consume it (include/xrl_hip.h)
__global__ void __launch_bounds__(256) pack_mid_frags16_kernel(const float* __restrict__ W, unsigned short* __restrict__ img) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * BH * BH; i += gridDim.x * blockDim.x) {
        const int n = i >> 7, k = i & (BH - 1);
        unsigned short h, m, l;
        split3(W[i], h, m, l);
        const int f = xrl_frag16_fwd_index(n, k), b = xrl_frag16_bwd_index(n, k);
        img[f] = h; img[XRL_FRAG16_PLANE + f] = m; img[2 * XRL_FRAG16_PLANE + f] = l;
        img[b] = h; img[XRL_FRAG16_PLANE + b] = m; img[2 * XRL_FRAG16_PLANE + b] = l;
    }
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_pack_mid_frags16(const xrl_ppo_fused_t* pp, uint16_t* image, int64_t image_elems, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr && image != nullptr && pp->params != nullptr && pp->n_layers >= 2);
    const xrl_fused_layer_t& L = pp->layers[1];
    XRL_CHECK_ARG(L.N == 2 * BH && L.K == BH && image_elems >= 3 * (int64_t)XRL_FRAG16_PLANE);
    hipLaunchKernelGGL(pack_mid_frags16_kernel, dim3(64), dim3(256), 0, as_stream(stream), pp->params + L.w_off, image);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

template <int ACT>
static int launch_bx_fwd(const xrl_ppo_fused_t& p, const xrl_sample_t& smp, hipStream_t stream) {
    const int n_tiles = (p.M + BPT - 1) / BPT;
    constexpr int LDSG = BxLds<BDMAX, BAMAX>::BYTES;
    if (p.dist == 0) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, false, false, 0, 0, 0, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSG, stream, p, smp);
    else if (p.out_act == XRL_ACT_TANH) hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, false, false, 0, 0, 2, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSG, stream, p, smp);
    else hipLaunchKernelGGL((ppo_trunk_bx_kernel<ACT, true, false, false, false, 0, 0, 1, true>), dim3(2 * n_tiles), dim3(FUSED_THREADS), LDSG, stream, p, smp);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

namespace xrl {
template <int ACT>
static int init_bx_fwd() {
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, false, false, 0, 0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<BDMAX, BAMAX>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, false, false, 0, 0, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<BDMAX, BAMAX>::BYTES));
    XRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_trunk_bx_kernel<ACT, true, false, false, false, 0, 0, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BxLds<BDMAX, BAMAX>::BYTES));
    return XRL_OK;
}
int init_ppo_trunk_bx_fwd() {
    if (int rc = init_bx_fwd<XRL_ACT_RELU>()) return rc;
    if (int rc = init_bx_fwd<XRL_ACT_LEAKY_RELU>()) return rc;
    if (int rc = init_bx_fwd<XRL_ACT_TANH>()) return rc;
    return XRL_OK;
}
}  // namespace xrl

extern "C" int xrl_trunk_forward16(const xrl_ppo_fused_t* pp, const xrl_sample_t* sample, xrl_stream_t stream) {
    XRL_CHECK_ARG(pp != nullptr);
    const xrl_ppo_fused_t& p = *pp;
    xrl_sample_t smp{};
    if (sample) {
        smp = *sample;
        XRL_CHECK_ARG(smp.n > 0 && smp.n <= p.M && smp.A == p.A && (p.M == smp.n || p.M == 2 * smp.n));
        XRL_CHECK_ARG(smp.act_out == nullptr || smp.gaussian == p.dist);      // (bootstrap-only calls carry no distribution)
        XRL_CHECK_ARG(smp.act_out == nullptr || smp.logp_out != nullptr);
        XRL_CHECK_ARG(smp.act_out == nullptr || !smp.gaussian || smp.log_std != nullptr);
        XRL_CHECK_ARG(smp.bootv_prev == nullptr || p.M == 2 * smp.n);
    }
    XRL_CHECK_ARG(p.params && p.frag16 && p.f_obs && p.M > 0 && (sample || (p.fwd_out && p.fwd_ld >= p.A + 1)));
    XRL_CHECK_ARG(p.D >= 1 && p.D <= BDMAX && p.A >= 1 && p.A <= BAMAX && (p.dist == 0 || p.dist == 1));
    XRL_CHECK_ARG(p.dist == 0 || p.out_act == XRL_ACT_NONE || p.out_act == XRL_ACT_TANH);
    XRL_CHECK_ARG(p.n_layers == 4 && p.n_head_layers == 2);
    const xrl_fused_layer_t &L0 = p.layers[0], &L1 = p.layers[1], &La = p.layers[2], &Lc = p.layers[3];
    XRL_CHECK_ARG(L0.K == p.D && L0.N == BH && L1.K == BH && L1.N == 2 * BH && La.K == BH && La.N == p.A && Lc.K == BH && Lc.N == 1);
    XRL_CHECK_ARG(La.in_off == 0 && Lc.in_off == BH && L0.act == L1.act);
    XRL_CHECK_ARG(L0.act == XRL_ACT_RELU || L0.act == XRL_ACT_LEAKY_RELU || L0.act == XRL_ACT_TANH);
    switch (L0.act) {
        case XRL_ACT_RELU: return launch_bx_fwd<XRL_ACT_RELU>(p, smp, as_stream(stream));
        case XRL_ACT_LEAKY_RELU: return launch_bx_fwd<XRL_ACT_LEAKY_RELU>(p, smp, as_stream(stream));
        default: return launch_bx_fwd<XRL_ACT_TANH>(p, smp, as_stream(stream));
    }
}

extern "C" int xrl_set_split_product_tr(int on) {
    g_bx_tr = on ? 1 : 0;
    return XRL_OK;
}

extern "C" int xrl_set_split_product_ksplit(int mode) {
    XRL_CHECK_ARG((mode >= 0 && mode <= 2) || mode == 5);            // (5: mode 1 with the barrier behind the small gradients moved behind the matrix phases)
    g_bx_ks = mode;
    return XRL_OK;
}
