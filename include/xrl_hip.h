/*
 * xrl_hip.h -- C ABI of libxrl_hip.so, the MI355X (gfx950) engine for XuanCe's RL update loop.
 *
 * The reference (agi-brain/xuance v1.4.4, /root/reference) is 100% Python and has no FFI; its plugin
 * seam for this path is the Python registries (xuance/torch/learners/__init__.py:41-99, Agent._build_memory
 * xuance/torch/agents/core/on_policy.py:65-104, Agent._build_learner xuance/torch/agents/base/agent.py:340).
 * The host-side classes in xuance_amd/ mirror those Python interfaces and bind the entry points below with
 * ctypes (INTEGRATION.md shows the binding a XuanCe maintainer would add).  Each entry point names the
 * reference code whose arithmetic it replaces.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch-ROCm allocations, passed as
 *     tensor.data_ptr()); the library never allocates or frees caller memory and keeps no hidden state
 *     besides the handles created by xrl_graph_* ;
 *   - all work is enqueued on the hipStream_t given (xrl_stream_t == hipStream_t); no hidden synchronisation,
 *     so every entry point is legal inside hipStreamBeginCapture (hipGraph capture);
 *   - return 0 on success, a negative XRL_E* code otherwise (never throws); xrl_last_error() gives the text;
 *   - all floating-point data is IEEE fp32 unless the name says otherwise; "f32 actions" follow the
 *     reference's buffers, which store action indices as float32 (memory_tools.py:12-41).
 *   - rollout data is a structure of arrays, TIME-MAJOR:  field[t][env][row]  (the reference is env-major
 *     [env][t][row], memory_tools.py:12-41, which makes the per-step write strided).
 */
#ifndef XRL_HIP_H
#define XRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* xrl_stream_t; /* hipStream_t */

enum { XRL_OK = 0, XRL_EINVAL = -1, XRL_EHIP = -2, XRL_EUNSUPPORTED = -3 };
enum { XRL_ACT_NONE = 0, XRL_ACT_RELU = 1, XRL_ACT_LEAKY_RELU = 2, XRL_ACT_TANH = 3, XRL_ACT_SIGMOID = 4 };

const char* xrl_version(void);
const char* xrl_last_error(void);
/* multiProcessorCount, warpSize and gcnArchName of the current device */
int xrl_device_info(int* cu_count, int* wave_size, char* arch, int arch_len);
/* one-time per-process device setup (kernel attributes); call once after the device is selected, outside capture */
int xrl_init(void);

/* ------------------------------------------------------------------ rollout buffer (structure of arrays) */

/* One field of a per-step write / per-batch gather.  row_bytes must be a multiple of 4. */
typedef struct {
    void* dst;        /* store: field base [T][n_envs][row_bytes]   gather: dense output [bs][row_bytes] */
    const void* src;  /* store: dense step data [n_envs][row_bytes]  gather: field base                  */
    int32_t row_bytes;
    int32_t flags;    /* gather only: bit0 = normalise this (scalar f32) field with stats (adv-norm)     */
} xrl_field_t;

/* store_element x6 (memory_tools.py:44-61,232-240; off-policy :365-372): field[t] <- src for every field.
 * Time-major layout makes each field one contiguous copy of n_envs*row_bytes bytes. fields: HOST array. */
int xrl_soa_store_step(const xrl_field_t* fields, int n_fields, int n_envs, int t, xrl_stream_t stream);
/* ... and *size_dev <- new_size in the same launch (the filled-slot count of a replay ring that device-side sampling follows) */
int xrl_soa_store_step_sized(const xrl_field_t* fields, int n_fields, int n_envs, int t, int32_t* size_dev, int32_t new_size,
                             xrl_stream_t stream);
/* The same store with the ring slot taken from a device counter (a captured vector step: the arguments do not change from step to
 * step): c = *counter_dev + offset; slot = (slot_bias + c) mod n_size; *size_dev = min(size_bias + c + 1, n_size).  The counter
 * is only read (the caller advances it with a launch of its own). */
int xrl_soa_store_step_ring(const xrl_field_t* fields, int n_fields, int n_envs, int n_size, int64_t slot_bias, int64_t size_bias,
                            const int32_t* counter_dev, int32_t offset, int32_t* size_dev, xrl_stream_t stream);

/* DummyOnPolicyBuffer.finish_path for every env and every closed path segment at once
 * (memory_tools.py:242-265, call sites ppo_agent.py:129-135,146-157).
 *   rew,val,term : [T][n_envs] f32     bootv : [T][n_envs] bootstrap value of a segment ending at (t,env)
 *   seg          : [T][n_envs] u8, bit0 = a path ends after step t (finish_path was called with ptr==t+1),
 *                  bit1 = that call passed a Python float (float64 carry, see oracle/xrl_oracle.py
 *                  gae_finish_path), bit2 = the bootstrap value of that call is 0 whatever bootv holds (set by the
 *                  device rollouts for terminated envs: finish_path(0.0, i), ppo_agent.py:132); steps after the last
 *                  closed segment of an env are left untouched.
 *   adv,ret      : [T][n_envs] outputs.  gamma/lam as double = the Python floats of the config.
 * Bit-exact with the reference for use_gae=1. */
int xrl_gae_scan(const float* rew, const float* val, const float* term, const float* bootv,
                 const uint8_t* seg, float* adv, float* ret, int n_envs, int T,
                 double gamma, double lam, int use_gae, xrl_stream_t stream);

/* mean / population std (ddof=0) of adv[idx] for n_batches consecutive minibatches of size bs
 * (memory_tools.py:281-282).  idx: int64 flat ENV-MAJOR indices (env*T + t, memory_tools.py:270).
 * stats out: [n_batches][2] f32 (mean, std).  Accumulates in float64. */
int xrl_adv_stats(const float* adv_field, const int64_t* idx, int bs, int n_batches, int n_envs, int T,
                  float* stats, xrl_stream_t stream);

/* DummyOnPolicyBuffer.sample / DummyOffPolicyBuffer.sample gather (memory_tools.py:64-84,267-287,374-387):
 * dst[b] = field[t_b][env_b] for every field; idx as above (env-major flat index, env*T + t).
 * Fields flagged bit0 are written as (x - stats[0]) / (stats[1] + 1e-8).  fields: HOST array. */
int xrl_soa_gather(const xrl_field_t* fields, int n_fields, const int64_t* idx, int bs, int n_envs, int T,
                   const float* stats, xrl_stream_t stream);
/* xrl_sample_replay_indices + xrl_soa_gather in ONE launch (bs <= 256): the rows are drawn inside the gather with the
 * same Philox stream, so both forms pick the same transitions; idx_out (optional, [bs]) receives the drawn flat indices. */
int xrl_soa_gather_sampled(const xrl_field_t* fields, int n_fields, int64_t* idx_out, int bs, int n_envs, int n_size,
                           const int32_t* size_dev, uint64_t seed, uint32_t counter, const uint32_t* counter_dev,
                           xrl_stream_t stream);

/* ------------------------------------------------------------------ dense layers on fp32 MFMA */

/* One GEMM of a grouped launch (all groups of a launch share the kernel, blockIdx.z selects the group). */
typedef struct {
    const float* A;    /* see each entry point */
    const float* B;
    float* C;
    const float* bias; /* fwd: [N] or NULL.  bwd_weight: unused */
    float* dbias;      /* bwd_weight: [n_split][...] slab address of the bias gradient or NULL */
    const float* aux;  /* bwd_data: activation OUTPUT of the layer that produced this layer's input, or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldaux;
    int32_t act;       /* fwd: activation applied to C.  bwd_data: activation whose derivative (from aux) scales C */
    int32_t pad;
} xrl_gemm_t;

/* nn.Linear (+activation) forward, mlp_block (rl_models/modules/layers.py:16-33):
 *   C[M,N] = act(A[M,K] . B[N,K]^T + bias[N])          A = input rows, B = weight [out,in] */
/*   Split-K (optional, per group): `aux` = caller-owned workspace of ldaux * M * N floats, ldaux = number of K ranges
 *   (2..64).  Each range is reduced by its own workgroups into its workspace slab and an epilogue launch adds the slabs
 *   in order and applies bias + activation: for layers with few output tiles and a long reduction (the 3 200 x 64 x 512
 *   convolution GEMMs of the Atari Q-network would otherwise occupy 50 of 256 CUs). */
int xrl_linear_fwd(const xrl_gemm_t* groups, int n_groups, xrl_stream_t stream);
/* xrl_linear_fwd without its split-K epilogue: groups that carry a workspace (aux, ldaux = K ranges > 1) leave their raw partial
 * sums ws[s][m][n] there and write nothing to C -- the caller's next launch sums them (xrl_ppo_act_tail). */
int xrl_linear_fwd_partials(const xrl_gemm_t* groups, int n_groups, xrl_stream_t stream);
/* backward w.r.t. the layer input (autograd of the same block):
 *   C[M,N] = (A[M,K] . B[K,N]) * act'(aux[M,N])        A = dY, B = weight [out=K,in=N] */
int xrl_linear_bwd_data(const xrl_gemm_t* groups, int n_groups, xrl_stream_t stream);
/* backward w.r.t. weight and bias, split over the batch into n_split deterministic partial slabs:
 *   C_s[N,K] = sum_{m in chunk s} A[m,N]^T X[m,K],  dbias_s[N] = sum_m A[m,N]     A = dY, B = X = layer input
 *   slab s of C is at C + s*slab_stride (floats); same for dbias. */
int xrl_linear_bwd_weight(const xrl_gemm_t* groups, int n_groups, int n_split, int64_t slab_stride,
                          xrl_stream_t stream);

/* ------------------------------------------------------------------ convolution stack of the Atari Q-network
 * (rl_models/representations/cnn.py:11-50, modules/layers.py:36-65) as im2col + the GEMMs above, NHWC end to end:
 *   y[B*OH*OW][F] = relu(col . W^T + b) with col = im2col(x), column order (c, kh, kw) == the reference weight
 *   layout [F][C][kh][kw]; OH = (H + 2p - k)/s + 1.  x: uint8 (scaled by /255.0 like cnn.py:45) or float32 NHWC. */
int xrl_im2col_nhwc(const void* x, int x_is_u8, float* col, int B, int H, int W, int C, int k, int s, int p,
                    xrl_stream_t stream);
/* dx[B][H][W][C] = col2im(dcol) (gather form, fixed summation order) * [xact > 0] (xact NULL: no mask) */
int xrl_col2im_nhwc(const float* dcol, const float* xact, float* dx, int B, int H, int W, int C, int k, int s, int p,
                    xrl_stream_t stream);
/* AdaptiveMaxPool2d((1,1)) over the P positions of y[B][P][F] -> feat[B][ld_feat], argmax[B][F] (NULL: not kept);
 * backward: dy[b][q][f] = dfeat[b][f] if q == argmax[b][f] and y[b][q][f] > 0 (the ReLU in front of the pool) else 0 */
int xrl_maxpool_hw_fwd(const float* y, float* feat, int32_t* argmax, int B, int P, int F, int ld_feat, xrl_stream_t stream);
int xrl_maxpool_hw_bwd(const float* dfeat, const int32_t* argmax, const float* y, float* dy, int B, int P, int F, int ld_dfeat,
                       xrl_stream_t stream);
/* nn.Flatten() of an NCHW activation (AC_CNN_Atari, rl_models/representations/cnn.py:90, configs/ppo/atari.yaml): the NHWC conv
 * output y[B][P][F] -> feat[b][f * P + q]; backward: dy[b][q][f] = dfeat[b][f * P + q] where y > 0 (the ReLU in front), else 0 */
int xrl_flatten_chw_fwd(const float* y, float* feat, int B, int P, int F, int ld_feat, xrl_stream_t stream);
int xrl_flatten_chw_bwd(const float* dfeat, const float* y, float* dy, int B, int P, int F, int ld_dfeat, xrl_stream_t stream);

/* ------------------------------------------------------------------ the same convolutions as IMPLICIT GEMMs on the fp32
 * matrix cores (csrc/conv_mfma.hip): no column matrix in memory.  One group = one convolution-shaped product
 *   out[b][hh*so + ph][ww*so + pw][n] = epi( sum_{th, tw, c} img[b][hh*sh + off_h + th][ww*sh + off_w + tw][c] * w[n][(th, tw, c)] )
 * over (hh, ww) in [0, nh) x [0, nw), taps Th x Tw, out-of-image taps read as 0; epi = activation(. + bias[n]) and, when `mask`
 * is given, times [mask[same index as out] > 0].  Three uses:
 *   forward of Conv2d(k, s, pad p) + ReLU (cnn.py:11-50, layers.py:36-65): img = layer input (uint8: values / 255.0 as cnn.py:45,
 *     or float32), Th = Tw = k, sh = s, off = -p, nh x nw = OH x OW, so = 1, ph = pw = 0;
 *   gradient w.r.t. the layer input: one group per residue class (rh, rw) of (h + p, w + p) mod s -- img = dY [B][OH][OW][F],
 *     taps = the kernel rows / columns of that class in reverse order, sh = 1, so = s, mask = the layer's input activation;
 *   weight gradient: xrl_conv_bwd_weight below.
 * `w` is NOT the reference tensor but an image of it in the order the matrix cores consume (a wave's 16-byte loads are
 * contiguous): with W'[n][(th, tw, c)] the reordered weight and the reduction index cut into groups of 32 and sub-steps of 4,
 *   w4[(((gq * 4 + s) * N/32 + nb) * 64 + lane) * 4 + j] = W'[nb*32 + lane%32][32 gq + 16 (lane/32) + 4 s + j];
 * xrl_gather_images builds the images from the flat parameters through index maps the host made once (nets.ConvStack).
 * Limits: N in {32, 64}, Th*Tw*C a multiple of 32, nw >= 2; C a power of two >= 16, or uint8 with C == 4 and Tw % 4 == 0. */
typedef struct {
    const void* img;
    const float* w;        /* fragment-ordered weight image (forward / input-gradient uses) */
    const float* bias;     /* [N] or NULL */
    const float* mask;     /* NULL, or [same shape as out] */
    float* out;            /* forward / input gradient: [B][OHt][OWt][N].  weight gradient: slab 0 of dW in the REFERENCE layout
                            * [N][C][Th][Tw] (slab s at out + s * slab_stride) */
    const float* dy;       /* weight gradient only: dY [B][nh][nw][N] */
    float* dbias;          /* weight gradient only: slab 0 of the bias gradient [N], or NULL */
    int32_t B, IH, IW, C;
    int32_t Th, Tw, nh, nw;
    int32_t sh, off_h, off_w, so;
    int32_t ph, pw, OHt, OWt;
    int32_t N, act, img_u8;
    int32_t pad;           /* weight gradient only: this group's own number of row chunks (<= n_split; 0: n_split) -- slabs beyond
                            * it are not written */
} xrl_conv_t;
/* k_split in {1, 2, 4}: waves of a workgroup that share one 32-row strip and split its reduction (4 / k_split strips per
 * workgroup); groups of one launch share N, img_u8 and k_split (<= 8 groups). */
int xrl_conv_fwd(const xrl_conv_t* groups, int n_groups, int k_split, xrl_stream_t stream);
/* diagnostics: the same launch writing 8 int64 time stamps per workgroup (tools/probe_conv_phases.py) */
int xrl_conv_fwd_probe(const xrl_conv_t* groups, int n_groups, int k_split, long long* dbg, xrl_stream_t stream);
/* dW[n][c][th][tw] = sum_rows dY[row][n] * patch(row)[(th, tw, c)], dbias[n] = sum_rows dY[row][n]; rows split into n_split
 * chunks, chunk s written to slab s (fixed order inside a chunk: four waves, then row pairs). */
int xrl_conv_bwd_weight(const xrl_conv_t* groups, int n_groups, int n_split, int64_t slab_stride, xrl_stream_t stream);
/* Weight-gradient groups of different kinds (32 | 64 filters, uint8 | float32 input) in ONE launch (1) or one launch per kind (0, the
 * default: the one-launch form measured slower, profiles/r06_g_conv_overlap.json) -- same numbers either way (tested). */
int xrl_set_conv_dw_mixed(int enable);
int xrl_conv_bwd_weight_probe(const xrl_conv_t* groups, int n_groups, int n_split, int64_t slab_stride, long long* dbg,
                              xrl_stream_t stream);   /* diagnostics, as xrl_conv_fwd_probe (launch the groups of ONE kernel variant) */
/* dst[j] = map[j] >= 0 ? src[map[j]] : 0 for up to 8 (src, map, dst, n) jobs in one launch: derived weight layouts from the
 * flat parameter buffer (map: image index -> parameter index, int32 [n]). */
typedef struct {
    const float* src;
    const int32_t* map;
    float* dst;
    int64_t n;
} xrl_image_job_t;
int xrl_gather_images(const xrl_image_job_t* jobs, int n_jobs, xrl_stream_t stream);

/* ------------------------------------------------------------------ PPO-clip loss (ppo_learner.py:46-60,70) */

typedef struct {
    const float* out;      /* [M][ld_out] actor head output: logits (categorical) or mu after activation_action */
    const float* value;    /* [M][ld_v] critic output (column 0) */
    const float* actions;  /* categorical: [M] f32 action index; gaussian: [M][A] f32 */
    const float* adv;      /* [M] advantages (already normalised unless stats != NULL) */
    const float* stats;    /* NULL or [2] (mean, std): adv <- (adv-mean)/(std+1e-8) on the fly */
    const float* returns;  /* [M] */
    const float* old_logp; /* [M] */
    const float* log_std;  /* gaussian: [A] parameter; categorical: NULL */
    float* d_out;          /* [M][ld_out] d loss / d (pre-activation head output) */
    float* d_value;        /* [M][ld_v]  d loss / d value (column 0) */
    float* d_log_std;      /* gaussian: [n_blocks][slab_stride...] see n_split; NULL for categorical */
    float* diag;           /* NULL or [4][M]: log_prob, ratio, surrogate1, surrogate2 (callback tensors) */
    double* partials;      /* [n_split][8]: sum min(s1,s2), sum (v-ret)^2, sum entropy, sum v, n_clipped, 0,0,0 */
    int32_t M, A, ld_out, ld_v;
    int32_t out_act;       /* activation already applied to `out` (gaussian activation_action), XRL_ACT_* */
    int32_t n_split;       /* number of blocks == number of gradient slabs */
    int64_t slab_stride;   /* floats between consecutive slabs of d_log_std */
    float clip_range, vf_coef, ent_coef;
    int32_t mode;          /* 0: PPO-clip surrogate; 1: A2C actor term -(adv * log_prob).mean() (a2c_learner.py:47), no ratio:
                            * old_logp is not read, partials[0] = sum adv*log_prob, n_clipped = 0;
                            * 2: PG actor term -(returns * log_prob).mean(), no critic (pg_learner.py:40-45): adv, value,
                            * d_value, old_logp are not touched;
                            * 3: PPO-KL (ppokl_learner.py:46-60): -(ratio * adv).mean() + kl_coef * KL(new || old).mean(), the old
                            * log-prob and the KL from the OLD DISTRIBUTION's parameters below (old_logp is not read);
                            * partials[5] = sum of KL (categorical: per row; gaussian: per row and dimension) */
    int32_t pad_mode;
    const float* old_a;    /* mode 3: [M][A] old logits (any normalisation) / old mu */
    const float* old_b;    /* mode 3, gaussian: [M][A] old std */
    const double* kl_coef; /* mode 3: [2] in device memory: [0] the coefficient (advanced by xrl_ppokl_adapt), [1] scratch of that call */
} xrl_ppo_loss_t;

/* kl_coef schedule of PPOKL_Learner.update (ppokl_learner.py:62-66): kl = sum_i partials[i][5] / count; > 1.5 target: x 2,
 * < 0.5 target: / 2, clipped to [0.1, 20]; kl_out (NULL or [1]) receives kl.  Launch after the loss of the same update.
 * kl_coef: [2] doubles -- [0] is advanced, [1] receives the value [0] had on entry (what this update's loss was formed with:
 * the host logs actor-loss = -surrogate + kl_coef_used * kl from it, also at the end of a chain of captured updates). */
int xrl_ppokl_adapt(const double* partials, int n_split, double count, double* kl_coef, double target_kl, float* kl_out,
                    xrl_stream_t stream);
int xrl_ppo_loss_categorical(const xrl_ppo_loss_t* p, xrl_stream_t stream);
int xrl_ppo_loss_gaussian(const xrl_ppo_loss_t* p, xrl_stream_t stream);
/* out[j] = sum_s partials[s][j]  (float64 in, float64 out), j < width */
int xrl_sum_partials(const double* partials, int n_rows, int width, double* out, xrl_stream_t stream);
/* n_batches independent sums in one launch: batch b reads partials + b*in_stride, writes out + b*out_stride (elements);
 * the loss terms of all updates of an update phase (n_epochs, e.g. off_policy_marl.py:341-343) leave with one launch. */
int xrl_sum_partials_batched(const double* partials, int n_rows, int width, double* out, int n_batches, long in_stride,
                             long out_stride, xrl_stream_t stream);

/* ------------------------------------------------------------------ optimiser
 * clip_grad_norm_ + torch.optim.Adam(eps=1e-5) + LinearLR.step (ppo_learner.py:18-22,61-67;
 * dqn_learner.py:18-22,47-53; marl_learner.py:64-75 with qmix_learner.py:88-96). */
typedef struct {
    int32_t step;          /* optimiser steps taken */
    int32_t sched_steps;   /* LinearLR steps taken */
    int32_t total_iters;   /* LinearLR total_iters */
    int32_t ticket;        /* internal: block-completion counter of xrl_adam_step, must start at 0 */
    double base_lr, end_factor, beta1, beta2, eps, weight_decay;
    double last_lr;        /* lr after the last scheduler step (what the reference logs) */
    double last_grad_norm; /* total norm before clipping */
} xrl_adam_state_t;        /* lives in DEVICE memory so a captured graph can advance it */

/* grad[p] = sum_s slabs[s][p]; sumsq_part[b] = sum over block b of grad^2 (float64). n_part blocks. */
/* xrl_grad_reduce with a fold region: slab columns [fold_off, fold_off + fold_len) (fold_off >= P) are added onto columns
 * [0, fold_len) after the main columns, in the same fixed slab order (see xrl_mirrors_t.fold_off). */
int xrl_grad_reduce_fold(const float* slabs, int n_split, int64_t slab_stride, int64_t P, float* grad, double* sumsq_part,
                         int n_part, int64_t fold_off, int fold_len, xrl_stream_t stream);
int xrl_grad_reduce(const float* slabs, int n_split, int64_t slab_stride, int64_t P, float* grad,
                    double* sumsq_part, int n_part, xrl_stream_t stream);
/* total_norm = sqrt(sum sumsq_part); grad *= min(1, max_norm/(total_norm+1e-6)) when max_norm > 0;
 * Adam update of params/m/v; advances *state (step, sched_steps, last_lr, last_grad_norm). */
int xrl_adam_step(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state,
                  const double* sumsq_part, int n_part, double max_norm, xrl_stream_t stream);
/* Same, and every updated parameter i is also written to dst_a[map_a[i]] / dst_b[map_b[i]] (map < 0: skip): keeps the
 * derived layouts of the fused kernels (transposed middle weights, packed LDS-cache image) current without extra
 * launches.  Maps are int32 [P] device arrays built once by the host. */
int xrl_adam_step_mirrored(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state,
                           const double* sumsq_part, int n_part, double max_norm, const int32_t* map_a, float* dst_a,
                           const int32_t* map_b, float* dst_b, xrl_stream_t stream);
/* General form: up to XRL_MAX_MIRRORS (map, dst) pairs (the fused kernels' fragment-ordered weight copies need two more). */
#define XRL_MAX_MIRRORS 4
typedef struct {
    const int32_t* map[XRL_MAX_MIRRORS];
    float* dst[XRL_MAX_MIRRORS];
    int32_t n;
    int32_t target_every;   /* > 0 with `target`: hard target-network update inside the optimiser launch -- when the step this
                             * launch performs is a multiple of target_every, target[i] <- new parameter
                             * (copy_target(), dqn_learner.py:56-57, qmix_learner.py:105-106; same as xrl_sync_target) */
    float* target;          /* NULL or [P] */
    float* target_image;    /* NULL, or a derived layout of the TARGET: whenever target[i] is refreshed, so is
                             * target_image[map[0][i]] (map[0][i] < 0: skip) */
    int64_t fold_off;       /* xrl_reduce_adam only: slab columns [fold_off, fold_off + fold_len) are a second partial of */
    int32_t fold_len;       /* columns [0, fold_len) (ppo_split_kernel's critic-role first-layer gradient); 0 = none */
    int32_t split_plane;    /* SPLIT mirror stores: a map value j <= -2 names 16-bit element e = -(j + 2) of dst (read as uint16_t*): the
                             * new parameter w goes there as three bf16 parts, dst16[e] = bf16(w), dst16[e + split_plane] = bf16(w - h),
                             * dst16[e + 2 split_plane] = bf16(w - h - m) (xrl_pack_mid_frags16's planes).  0 with no such map. */
    /* xrl_reduce_adam[_exchange] only -- two bookkeeping launches of an update phase riding in the optimiser launch (one of
     * its blocks does them next to its slab loads; at the DQN-C3 update they were 8 of 150 us as launches of their own): */
    uint32_t* tick;         /* NULL, or a device counter advanced by tick_inc (xrl_counter_add: the replay draw counter) */
    const double* part;     /* NULL, or [part_rows][8] float64 loss partials ... */
    double* part_out;       /* ... summed row by row into part_out[8] (xrl_sum_partials, same order) */
    int32_t tick_inc, part_rows;
    /* xrl_reduce_adam[_exchange] only: parameter ranges [alt_lo[i], alt_hi[i]) (multiples of 4) whose gradient is the sum of the
     * first alt_split slab rows instead of n_split (xrl_wide_dw1's parts); alt_split = 0: none */
    int64_t alt_lo[2], alt_hi[2];
    int32_t alt_split, pad2;
} xrl_mirrors_t;
int xrl_adam_step_mirrors(float* params, float* grad, float* m, float* v, int64_t P, xrl_adam_state_t* state,
                          const double* sumsq_part, int n_part, double max_norm, const xrl_mirrors_t* mirrors,
                          xrl_stream_t stream);
/* xrl_grad_reduce + xrl_adam_step_mirrors in ONE launch (same numbers): the blocks meet at a counter barrier between
 * the slab reduction and the Adam update (all of them are resident), the reduced gradient never leaves the chip in
 * between.  Needs P % 4 == 0, 16-byte aligned slabs, n_part >= ceil(P/256); sync: [4 + ceil(P/256)] uint32 scratch, zero-initialised
 * once (sync[2] != 0 afterwards reports a barrier time-out; the update of that call is then invalid).
 * max_norm <= 0 (use_grad_clip: False -- the QMIX and DQN configs): nothing depends on the total norm, so the blocks do
 * NOT meet: each goes from its slab sums straight to its Adam quads, and the last block out adds up sumsq_part for
 * state->last_grad_norm and advances *state. */
int xrl_reduce_adam(const float* slabs, int n_split, int64_t slab_stride, float* params, float* grad, float* m, float* v,
                    int64_t P, xrl_adam_state_t* state, double* sumsq_part, int n_part, double max_norm,
                    const xrl_mirrors_t* mirrors, uint32_t* sync, xrl_stream_t stream);
/* 1 if xrl_reduce_adam (with_exchange: xrl_reduce_adam_exchange) can run for P parameters on the current device: P % 4 == 0, at
 * most 1 024 groups of 256, and every block of the launch resident at once (hipOccupancyMaxActiveBlocksPerMultiprocessor x the
 * compute units: the launch's inter-block barrier spins).  Callers fall back to xrl_grad_reduce + xrl_adam_step otherwise. */
int xrl_reduce_adam_fits(int64_t P, int with_exchange);

/* Data-parallel ranks (one process per GPU; replaces the DistributedDataParallel gradient all-reduce of
 * xuance/torch/learners/learner.py:60-65 + torch's DDP hooks for this path): xrl_reduce_adam in which the ranks average
 * their reduced gradients INSIDE the launch, through exchange buffers every rank allocates with xrl_ipc_alloc and maps
 * from its peers with xrl_ipc_open (xGMI peer access; two ranks on one GPU work the same way).  Averaging is the fp32 sum
 * in rank order times 1/world on every rank, so the replicas stay bit-identical; norm, clip and Adam then act on the
 * averaged gradient as in xrl_reduce_adam.  A peer that does not show up within max_spins polls sets sync[2] = 2 and makes
 * the norm NaN (the update of that call is invalid).  With clipping on, the launch's own inter-block barrier (the norm) is reached by a
 * block only after its peer wait, so its bound is 2 000 000 + 4 * max_spins polls here (2 000 000 in xrl_reduce_adam).
 * Buffer layout: uint32 flags[2][XRL_XC_MAX_GROUPS] | float data[2][4 * stride4]  (index 0/1 = parity of the step). */
#define XRL_XC_MAX_RANKS 8
#define XRL_XC_MAX_GROUPS 1024
#define XRL_XC_DATA_OFFSET (2 * XRL_XC_MAX_GROUPS)      /* in 4-byte words from the buffer base */
#define XRL_IPC_HANDLE_BYTES 64
typedef struct {
    float* base[XRL_XC_MAX_RANKS];   /* base[rank] = this rank's buffer, base[r] = rank r's buffer as mapped here */
    int64_t stride4;                 /* float4 slots per parity (>= P / 4) */
    int32_t world, rank;
    int32_t max_spins, pad;
    float inv_world, pad2;
} xrl_exchange_t;
#define XRL_XC_BYTES(stride4) (4 * (size_t)XRL_XC_DATA_OFFSET + 2 * 16 * (size_t)(stride4))
int xrl_reduce_adam_exchange(const float* slabs, int n_split, int64_t slab_stride, float* params, float* grad, float* m,
                             float* v, int64_t P, xrl_adam_state_t* state, double* sumsq_part, int n_part, double max_norm,
                             const xrl_mirrors_t* mirrors, uint32_t* sync, const xrl_exchange_t* exchange,
                             xrl_stream_t stream);
/* Exchange-buffer memory: zero-filled fine-grained device memory + its 64-byte IPC handle; open / close a peer's handle;
 * free one's own; clear (stream-ordered memset, outside graph capture only). */
int xrl_ipc_alloc(size_t bytes, void** ptr_out, unsigned char* handle_out);
int xrl_ipc_open(const unsigned char* handle, void** ptr_out);
int xrl_ipc_close(void* peer_ptr);
int xrl_ipc_free(void* ptr);
int xrl_ipc_clear(void* ptr, size_t bytes, xrl_stream_t stream);


/* ------------------------------------------------------------------ rollout-side ops (one small launch per step)
 * RunningMeanStd.update + _process_observation (statistic_tools.py:117-185, agent.py:262-283) */
typedef struct {
    const float* x;       // [n][D] raw observations
    float* mean;          // [D] running mean (float32 like the reference's np.float32 arrays)
    float* var;           // [D]
    double* count;        // [1] python float in the reference
    float* out0;          // normalised copy 0 (e.g. the policy input rows), may be NULL
    float* out1;          // normalised copy 1 (e.g. the rollout-buffer slot of this step), may be NULL
    int n, D, ld_x, ld0, ld1;
    int update;           // 1: RunningMeanStd.update(x) first (ppo_agent.py:114)
    int normalize;        // 1: clip((x-mean)/(std+1e-8), +-range) (agent.py:262-283); 0: plain copy
    float range;
} xrl_rms_t;
int xrl_obs_normalize(const xrl_rms_t* p, xrl_stream_t stream);

/* OnPolicyAgent.get_actions (core/on_policy.py:128-169): sample an action from the policy head output, its
 * log-prob and the value; inverse-CDF / Box-Muller on Philox4x32-10 randomness, or on supplied noise. */
typedef struct {
    const float* heads;     // [2n or n][ld] policy head buffer: cols [0,A) actor output, col A value
    const float* log_std;   // gaussian: [A]
    const float* noise;     // optional supplied randomness (parity tests): categorical [n] uniforms, gaussian [n][A] normals
    float* act_out;         // rollout-buffer slot actions[t]: [n] (categorical, f32 index) or [n][A]; NULL = only bootv_prev
    float* val_out;         // values[t] [n]
    float* logp_out;        // old_logp[t] [n]
    int32_t* env_action;    // discrete action for the device env [n] (categorical) or NULL
    float* env_action_f;    // continuous action for the device env [n][A] or NULL
    float* bootv_prev;      // NULL or bootv[t-1] [n]: value of the previous step's (normalised) next_obs = rows [n,2n)
    int n, A, ld, gaussian;
    uint64_t seed;
    uint32_t step;          // global vector-step counter (RNG counter)
    const uint32_t* step_dev; // if non-NULL the counter is read from device memory (graph replay) and `step` is added
} xrl_sample_t;
int xrl_policy_sample(const xrl_sample_t* p, xrl_stream_t stream);

/* Device-resident CartPole-v1 vector env with the DummyVecEnv auto-reset contract
 * (environment/vector_envs/dummy_vec_env.py:65-76: terminal obs returned, reset obs becomes buf_obs). */
typedef struct {
    double* state;            // [n][4] x, x_dot, theta, theta_dot (float64 like Gymnasium)
    int32_t* steps;           // [n] elapsed steps of the running episode
    int32_t* episodes;        // [n] finished-episode counter (also the reset RNG counter)
    const int32_t* action;    // [n] 0/1
    float* obs;               // [n][4] observation the agent sees next (after auto-reset)      == buf_obs
    float* next_obs;          // [n][4] observation returned by step() before any reset (terminal obs on episode end)
    float* reward;            // [n]
    float* terminated;        // [n] 0/1
    float* truncated;         // [n] 0/1
    float* ep_score;          // [n] running episode score (XuanCeEnvWrapper.episode_score, utils/wrapper.py)
    double* stats;            // [4] totals: finished episodes, sum of scores, sum of lengths, (unused)
    int n, max_steps;
    uint64_t seed;
} xrl_cartpole_t;
int xrl_cartpole_step(const xrl_cartpole_t* p, int reset, xrl_stream_t stream);

/* Device-resident Pendulum-v1 (kind 1), MountainCar-v0 (2), Acrobot-v1 (3): the other environments of the reference's
 * the yaml files of configs/ppo/classic_control, same contract as xrl_cartpole_step.  Dynamics as published with Gymnasium's classic_control
 * package (third-party; csrc/classic.h restates them, oracle/xrl_oracle.py is the NumPy twin).  obs: [n][3 | 2 | 6]. */
typedef struct {
    double* state;            /* [n][4] (Pendulum: theta, theta_dot; MountainCar: position, velocity; Acrobot: all four) */
    int32_t* steps;           /* [n] */
    int32_t* episodes;        /* [n] finished-episode counter (also the reset RNG counter) */
    const int32_t* action;    /* [n] discrete action (MountainCar, Acrobot: 0 / 1 / 2) or NULL */
    const float* action_f;    /* [n] continuous action (Pendulum: torque, clipped to +-2 here) or NULL */
    float* obs;               /* [n][D] observation the agent sees next (after auto-reset) */
    float* next_obs;          /* [n][D] observation returned by the step before any reset */
    float* reward; float* terminated; float* truncated;   /* [n] */
    float* ep_score;          /* [n] */
    double* stats;            /* [4] finished episodes, sum of scores, sum of lengths, - */
    int32_t n, kind, max_steps, pad;
    uint64_t seed;
} xrl_classic_t;
int xrl_classic_step(const xrl_classic_t* p, int reset, xrl_stream_t stream);

/* The tail of an on-policy acting step on a device env as ONE launch (round 6; csrc/act_tail.hip): the heads' product
 * (actor_head.py / critic_head.py: A outputs on the actor branch's features, one value on the critic's), OnPolicyAgent.get_actions'
 * sampling (on_policy.py:128-169 = xrl_policy_sample) and the env's step (xrl_classic_step / xrl_cartpole_step) -- what a vector step
 * of PPO_Agent's general path does in three launches after the hidden layers.  Rows [0, n) of hb are this step's observations, rows
 * [n, 2n) the previous step's next observations (their value -> sample.bootv_prev).  Bit-identical to xrl_linear_fwd (heads) +
 * xrl_policy_sample + the env step.  K <= 128 and a multiple of 32, A <= 8, both feature blocks inside the first 256 columns. */
typedef struct {
    const float* hb;            /* [2n | n][ldh] the level the heads read */
    const float* w_actor;       /* [A][ldw_a], k contiguous */
    const float* b_actor;       /* [A] or NULL */
    const float* w_critic;      /* [1][ldw_c] */
    const float* b_critic;      /* [1] or NULL */
    float* heads;               /* NULL or [2n][A + 1] out: what the heads' xrl_linear_fwd writes (rows [n, 2n): the value column only,
                                 * unless boot_actor) */
    int32_t ldh, K, a_off, c_off;      /* actor features: columns [a_off, a_off + K) of a row of hb; critic: [c_off, c_off + K) */
    int32_t ldw_a, ldw_c;
    int32_t boot_rows;          /* 1: rows [n, 2n) exist (sample.bootv_prev may be set) */
    int32_t boot_actor;         /* 1: also the actor outputs of rows [n, 2n) (nobody reads them) */
    int32_t env_kind;           /* 0: no env step; 1..3: xrl_classic_t.kind (classic); 4: CartPole (cartpole) */
    int32_t act_actor;          /* XRL_ACT_* applied to the actor's outputs (activation_action of a Gaussian head), XRL_ACT_NONE else */
    xrl_sample_t sample;        /* .heads is ignored; n, A, ld = A + 1; act_out NULL: bootstrap values only (no env step) */
    xrl_classic_t classic;
    xrl_cartpole_t cartpole;
} xrl_act_tail_t;
int xrl_act_tail(const xrl_act_tail_t* p, xrl_stream_t stream);
/* Measurement aid (tools/probe_act_tail.py): stamps != NULL -> workgroup 0 of the following xrl_act_tail launches writes its s_memtime at
 * start / features staged / heads done / sampled / end to stamps[0..4]; NULL switches it off.  Not for production use. */
int xrl_debug_act_tail_stamps(long long* stamps);

/* Synthetic MuJoCo-shaped vector env on the device (an input provider for the continuous-control shapes of BASELINE
 * config C4 -- no simulator is installed; NOT a reference component): state' = tanh(state.A + clip(a).B) + 0.01 N(0,1),
 * reward = state'[0] - 0.1 |a|^2, truncation after max_steps, same auto-reset contract as xrl_cartpole_step. */
typedef struct {
    float* state;             /* [n][D] */
    int32_t* steps;           /* [n] */
    const float* action;      /* [n][A] */
    const float* Amat;        /* [D][D] */
    const float* Bmat;        /* [A][D] */
    float* obs;               /* [n][D] observation the agent sees next (after auto-reset) */
    float* next_obs;          /* [n][D] pre-reset next observation */
    float* reward; float* terminated; float* truncated;   /* [n] */
    float* ep_score;          /* [n] */
    double* stats;            /* [4] finished episodes, sum of scores, sum of lengths, - */
    int32_t n, D, A, max_steps;
    uint64_t seed;
    uint32_t step; const uint32_t* step_dev;
} xrl_synth_ctl_t;
int xrl_synth_control_step(const xrl_synth_ctl_t* p, int reset, xrl_stream_t stream);

/* Atari-shaped input provider (no emulator in the image; configs/dqn/atari.yaml:7-8 for the 84x84 x 4-frame uint8 shape):
 * per step a fresh frame stack per env (next_obs), reward, episode end with probability p_term or at max_steps,
 * auto-reset.  cur_obs (what the agent acts on next) must be a different buffer from the one it acted on in this step:
 * the env alternates two, so (obs, next_obs) go to the replay ring without a copy.  Input provider, not measured path. */
typedef struct {
    uint8_t* cur_obs; uint8_t* next_obs;                     /* [n][row_bytes] */
    const int32_t* action;                                   /* [n] */
    float* reward; float* terminated; float* truncated; float* done;   /* [n] */
    int32_t* steps; int32_t* end_step;                       /* [n] */
    int32_t n, row_bytes, A, max_steps;
    float p_term, pad;
    uint64_t seed;
    uint32_t step; const uint32_t* step_dev;
} xrl_synth_frames_t;
int xrl_synth_frames_step(const xrl_synth_frames_t* p, int reset, xrl_stream_t stream);

/* SMAC-3m-shaped multi-agent input provider (no simulator in the image; docs/source/documents/benchmark/smac/smac.rst:15,19
 * for the agent count / episode limit): per step fresh observations [n][N*O], global state [n][S] and availability masks
 * [n][N*A] (action 0 always available), team reward, episode end with probability p_term or at max_steps, auto-reset
 * (next_* = what the step returned, buf_* = what the agents act on next).  Input provider, not part of the measured path. */
typedef struct {
    float* buf_obs; float* buf_state; float* buf_avail;      /* acted on next (post-reset) */
    float* next_obs; float* next_state; float* next_avail;   /* returned by this step (pre-reset) */
    const int32_t* action;                                   /* [n][N] */
    float* rewards; float* terminals;                        /* [n][N] */
    float* terminated; float* truncated; float* done;        /* [n] */
    int32_t* steps; int32_t* end_step;                       /* [n] running / final step count of the episode */
    int32_t n, N, O, S, A, max_steps;
    float p_term, pad;
    uint64_t seed;
    uint32_t step; const uint32_t* step_dev;
    const float* prev_state;   /* optional: the state the agents acted on, when buf_* are the other of two alternating buffers */
    int32_t* prev_steps;       /* optional [n]: step index of this transition inside its episode */
    int64_t* totals;           /* optional [2]: running totals of finished episodes and of the env steps in them */
} xrl_synth_marl_t;
int xrl_synth_marl_step(const xrl_synth_marl_t* p, int reset, xrl_stream_t stream);

/* Per-step bookkeeping of PPO_Agent.train (ppo_agent.py:128,144-157): reward normalisation + store,
 * path-end flags, return tracker and ret_rms updates in env order, normalised next_obs for bootstrapping. */
typedef struct {
    const float* reward;      // [n] raw env reward
    const float* terminated;  // [n]
    const float* truncated;   // [n]
    const float* next_obs;    // [n][D] raw pre-reset next observation
    const float* obs_mean;    // [D] observation running stats AFTER this step's update
    const float* obs_var;     // [D]
    float* next_obs_norm;     // [n][ld_next] normalised next_obs -> policy input rows [n,2n) of the next forward
    float* rew_out;           // rewards[t]   (normalised reward, agent.py:285-294)
    float* term_out;          // terminals[t]
    uint8_t* seg_out;         // seg[t]: path-end flags for xrl_gae_scan
    float* ret_track;         // [n] discounted return tracker (ppo_agent.py:144)
    float* ret_mean;          // [1] RunningMeanStd of returns
    float* ret_var;           // [1]
    double* ret_count;        // [1]
    int n, D, ld_next;
    int use_obsnorm, use_rewnorm, last_step;   // last_step: buffer becomes full -> every env closes its path
    float obs_range, rew_range, gamma;
    /* NULL, or [n]: what PG_Agent's get_terminated_values returns for env e at the moment its path closes (pg_agent.py:66-79:
     * _process_reward(rewards), i.e. the raw reward over the CURRENT return statistics): for an env whose episode ended in this step
     * that is after ret_rms.update of every finished env up to and including itself (on_policy.py:272-283); at last_step and for
     * every other env the value stored in rew_out */
    float* pg_bootv;
} xrl_poststep_t;
int xrl_rollout_poststep(const xrl_poststep_t* p, xrl_stream_t stream);
/* xrl_rollout_poststep(post) followed by xrl_obs_normalize(rms) as ONE launch (round 6): the previous vector step's bookkeeping and
 * this step's RunningMeanStd.update + normalisation, both single workgroups; same results as the two launches. */
int xrl_post_norm(const xrl_poststep_t* post, const xrl_rms_t* rms, xrl_stream_t stream);

/* The tail of an on-policy ACTING pass of an actor-critic network whose heads sit directly on one wide hidden layer (AC_CNN_Atari of
 * configs/ppo/atari.yaml: Linear(6 400, 512) + ReLU, CategoricalActorHead / ValueHead on the 512 features; cnn.py:53-102,
 * on_policy.py:128-169) as ONE launch (round 6): the split-K partial sums xrl_linear_fwd_partials left for that layer are summed
 * (xrl_linear_fwd's own epilogue: partials in order, + bias, activation), the actor's logits and the critic's value follow as one
 * wavefront per (row, head) with xrl_linear_fwd's skinny-layer arithmetic, rows [0, n) are sampled exactly as xrl_policy_sample does
 * (same Philox draws / supplied uniforms; action, log-prob, value -> the buffer slot, the env's action), rows [n, M) give bootv_prev --
 * the same numbers as the three launches it replaces, bit for bit.  Beside that workgroup: one workgroup does the PREVIOUS vector
 * step's bookkeeping (post_n > 0: xrl_rollout_poststep's arguments in `post`) and a few copy the observations the policy acted on
 * into their buffer slot (memory.observations[t] = obs, ppo_agent.py:128: copy_bytes > 0). */
typedef struct {
    const float* ws;            /* [ks][M][H] partial sums of the hidden layer's product (xrl_linear_fwd_partials) */
    const float* bias;          /* [H] */
    int32_t ks, M, H, act;      /* M = rows of the policy batch (<= 64), H % 64 == 0, H <= 1 024; act: XRL_ACT_* of the hidden layer */
    const float* w_actor; const float* b_actor;     /* [A][H], [A] */
    const float* w_critic; const float* b_critic;   /* [1][H], [1] */
    /* xrl_sample_t's fields (categorical) */
    const float* noise;
    float* act_out; float* val_out; float* logp_out; int32_t* env_action; float* bootv_prev;
    int32_t n, A;
    uint64_t seed; uint32_t step; uint32_t pad0; const uint32_t* step_dev;
    float* heads_out;           /* NULL, or [M][A + 1]: the head outputs (tests) */
    xrl_poststep_t post;        /* the previous step's bookkeeping ... */
    int32_t post_n, pad1;       /* ... when post_n > 0 */
    const void* copy_src; void* copy_dst; int64_t copy_bytes;     /* 16-byte aligned, copy_bytes % 16 == 0 */
} xrl_ppo_act_tail_t;
int xrl_ppo_act_tail(const xrl_ppo_act_tail_t* p, xrl_stream_t stream);

/* OffPolicyAgent.exploration (core/off_policy.py:129-148): greedy argmax + per-env epsilon coin. */
typedef struct {
    const float* q;          // [n][ld] Q-values
    const float* uniforms;   // optional [n] supplied uniforms
    const int32_t* randoms;  // optional [n] supplied random actions
    const float* eps_dev;    // epsilon in device memory, or NULL: `eps` below (eager loops: the host knows the value, no fill launch)
    int32_t* action;         // [n]
    float* action_f;         // [n] float copy for the replay buffer (actions are stored as float32)
    int n, A, ld;
    float eps;               // used when eps_dev is NULL
    uint64_t seed;
    uint32_t step;
    const uint32_t* step_dev;
} xrl_egreedy_t;
int xrl_egreedy(const xrl_egreedy_t* p, xrl_stream_t stream);
/* OffPolicyMARLAgents action selection (core/off_policy_marl.py:212-255,289-308; value_factorization.py:87-92):
 * greedy = argmax_a Q with unavailable actions at -1e10; exploration = ONE coin for the whole vector step, and when it
 * lands every agent of every env takes a uniformly random AVAILABLE action (Categorical(avail_mask).sample()). */
typedef struct {
    const float* q;          /* [R][ld] per-agent Q-values, R = n_envs * n_agents */
    const float* avail;      /* NULL or [R][A] f32 0/1 */
    const float* eps_dev;    /* [1] epsilon in device memory, or NULL: `eps` below */
    const float* coin;       /* NULL or [1] supplied uniform for the step coin (parity tests) */
    const float* uniforms;   /* NULL or [R] supplied uniforms for the random actions */
    int32_t* action;         /* [R] */
    float* action_f;         /* NULL or [R] float32 copy (the replay buffer stores actions as float32) */
    int32_t R, A, ld;
    float eps;               /* used when eps_dev is NULL */
    uint64_t seed;
    uint32_t step; const uint32_t* step_dev;
} xrl_marl_act_t;
int xrl_marl_select_actions(const xrl_marl_act_t* p, xrl_stream_t stream);

/* out[e][j] (j < take) = first `take` entries of the e-th of n_perm independent pseudo-random permutations of [0, N):
 * the minibatch indices of one update phase (np.random.shuffle per epoch, core/on_policy.py:194-204).  A keyed bijection
 * evaluated per element (no sort), keyed by (seed, counter + *counter_dev, e).  take == N gives full permutations. */
int xrl_random_permutation(int64_t* out, int n_perm, int64_t N, int64_t take, uint64_t seed, uint32_t counter,
                           const uint32_t* counter_dev, xrl_stream_t stream);

/* out[b] = env_b * n_size + step_b with env_b ~ U{0..n_envs-1}, step_b ~ U{0..*size_dev-1}: uniform replay sampling with
 * replacement (memory_tools.py:376-377, memory_tools_marl.py:753-754) on the device; *size_dev is the number of valid
 * ring rows, read at run time so a captured graph follows the filling buffer.  Keyed by (seed, counter + *counter_dev). */
int xrl_sample_replay_indices(int64_t* out, int bs, int n_envs, int n_size, const int32_t* size_dev, uint64_t seed,
                              uint32_t counter, const uint32_t* counter_dev, xrl_stream_t stream);

/* *counter += inc on the stream (advances RNG step counters between replays of a captured rollout). */
int xrl_counter_add(uint32_t* counter, uint32_t inc, xrl_stream_t stream);

/* ------------------------------------------------------------------ fused rollout step (ONE launch per vector step)
 * The whole body of the on-policy agent's step loop (ppo_agent.py:113-177) for a device-resident CartPole:
 * deferred ret_rms merges, obs_rms.update + normalisation + store, the complete actor-critic MLP forward on
 * LDS-resident 32-row tiles (fp32 MFMA), action sampling / log-prob / value + store, physics with auto-reset,
 * reward normalisation + store, path-end flags, return tracker, normalised next_obs for bootstrapping.
 * Workgroup i < n/32 owns envs [32i, 32i+32) ("act" tile); workgroups i >= n/32 evaluate V(next_obs of the previous
 * step) for the same envs ("boot" tile -> bootv[t-1]).  All cross-workgroup inputs are read from *_in buffers written
 * by the previous launch and all outputs go to *_out buffers (the host ping-pongs them), so no in-launch
 * synchronisation between workgroups is needed; statistics that couple all envs are recomputed redundantly per
 * workgroup from the (tiny) *_in arrays.  Produces the same numbers as the unfused sequence
 * xrl_obs_normalize -> xrl_linear_fwd x L -> xrl_policy_sample -> xrl_cartpole_step -> xrl_rollout_poststep. */
#define XRL_FUSED_MAX_LAYERS 8
#define XRL_FUSED_MAX_LEVELS 6
typedef struct {
    int32_t w_off, b_off;          /* float offsets of weight [N][K] and bias [N] in the flat parameter buffer */
    int32_t K, N, act;
    int32_t in_level, in_off, out_level, out_off;
    int32_t pad;
} xrl_fused_layer_t;

/* Whole feed-forward plans as ONE launch (round 6; csrc/mlp_chain.hip): every job is a DAG of Linear(+activation) layers on LEVELS of a
 * row tile (level 0 = the input, a layer reads columns [in_off, in_off + K) of an earlier level and writes [out_off, out_off + N) of a
 * later one: xrl_fused_layer_t, listed in execution order); up to XRL_CHAIN_MAX_JOBS independent jobs share the launch (an eval network
 * and its target twin, the mixers' hyper-networks).  Bit-identical to one xrl_linear_fwd per stage.  out[l] != NULL: level l of every
 * row is written to out[l][row * ld_out[l] + col].  tile0: prefix sums of ceil(M / 32) over the jobs (tile0[0] = 0). */
#define XRL_CHAIN_MAX_JOBS 4
typedef struct {
    const float* x;             /* [M][ldx] input rows; level_width[0] columns are read */
    const float* params;        /* flat parameter buffer the layers' offsets refer to */
    xrl_fused_layer_t layers[XRL_FUSED_MAX_LAYERS];
    int32_t n_layers, n_levels;
    int32_t level_width[XRL_FUSED_MAX_LEVELS];
    float* out[XRL_FUSED_MAX_LEVELS];
    int32_t ld_out[XRL_FUSED_MAX_LEVELS];
    int32_t ldx, M;
} xrl_mlp_chain_job_t;
typedef struct {
    xrl_mlp_chain_job_t job[XRL_CHAIN_MAX_JOBS];
    int32_t n_jobs;
    int32_t tile0[XRL_CHAIN_MAX_JOBS + 1];
    int32_t pad[2];
} xrl_mlp_chain_t;
int xrl_mlp_chain_fwd(const xrl_mlp_chain_t* p, xrl_stream_t stream);
int xrl_mlp_chain_lds_bytes(const xrl_mlp_chain_t* p);     /* LDS the launch needs (must be <= 160 KB), -1 on a malformed description */
/* Measurement aid (tools/probe_mlp_chain.py): stamps != NULL -> workgroup 0 of the following launches writes its s_memtime at start,
 * inputs staged, after every stage, after the write-back to stamps[0..], their count to stamps[15]; NULL: off. */
int xrl_debug_mlp_chain_stamps(long long* stamps);

typedef struct {
    const float* params;                               /* flat parameter buffer */
    const float* cache_image;                          /* packed LDS parameter-cache image (xrl_pack_rollout_cache) */
    const float* frag_image;                           /* NULL or the prefetched layer's weights in MFMA-fragment order */
    xrl_fused_layer_t layers[XRL_FUSED_MAX_LAYERS];
    int32_t n_layers, n_levels;
    int32_t n_head_layers, pad0;                       /* trailing layers that write the last (head) level */
    int32_t level_width[XRL_FUSED_MAX_LEVELS];        /* level 0 = observation */
    /* cross-workgroup state, ping-ponged by the host */
    const float* obs_raw_in;   float* obs_raw_out;     /* [n][D] raw observation the agent acts on */
    const float* xnext_in;     float* xnext_out;       /* [n][D] normalised next_obs (pre-reset) */
    const float* obs_stats_in; float* obs_stats_out;   /* [2*D] mean | var (float32) */
    const double* obs_count_in; double* obs_count_out; /* [1] */
    const float* ret_stats_in; float* ret_stats_out;   /* [2] mean, var */
    const double* ret_count_in; double* ret_count_out; /* [1] */
    const uint8_t* ended_in;   uint8_t* ended_out;     /* [n] episode ended at the previous / this step */
    const float* ret_final_in; float* ret_final_out;   /* [n] discounted return of an episode that just ended */
    float* ret_track;                                  /* [n] return tracker (owner workgroup only) */
    /* rollout-buffer slots of this step */
    float* obs_slot; float* act_slot; float* val_slot; float* logp_slot; float* rew_slot; float* term_slot;
    uint8_t* seg_slot; float* bootv_prev;              /* bootv_prev: bootv[t-1] or NULL (no boot tiles) */
    const float* log_std;                              /* gaussian head: [A] */
    /* CartPole state */
    double* cp_state; int32_t* cp_steps; int32_t* cp_episodes; float* cp_score; double* cp_stats;
    int32_t n, D, A, gaussian, max_steps;
    int32_t use_obsnorm, use_rewnorm, last_step, boot_only;
    int32_t role_split, split_col;                      /* actor / critic branches of the stacked layer in separate workgroups */
    float obs_range, rew_range, gamma, pad1;
    uint64_t seed, env_seed;
    uint32_t step; const uint32_t* step_dev;
    long long* dbg;                                    /* NULL, or [16] shader-clock stamps of the last workgroup (diagnostics) */
} xrl_rollout_step_t;
int xrl_rollout_step_cartpole(const xrl_rollout_step_t* p, xrl_stream_t stream);
/* ------------------------------------------------------------------ rollout of the 4-128-{128-2,128-1} class (csrc/rollout_actor.hip)
 * The step loop of PPO_Agent.train (ppo_agent.py:111-177) for a device-resident CartPole with ONLY the actor on the step
 * chain: xrl_rollout_cartpole_run executes vector steps [t0, t0 + n_steps) of a rollout of horizon T -- obs_rms.update,
 * normalisation, actor forward (fp32 MFMA, 16-row tiles), Categorical sample / log-prob, physics with auto-reset, path-end
 * flags, return tracker, normalised rewards and ret_rms.update in env order -- in ONE launch (n_steps = T: the whole
 * rollout; n_steps = 1: one launch per vector step, same arithmetic, bit-identical).  It leaves values / bootstrap values
 * to xrl_rollout_cartpole_values, which evaluates V(obs[t][e]) for the same steps and V(next_obs[t][e]) where seg == 1
 * (a path cut without termination: truncation / buffer full) as one batched pass over the whole device.
 * State is updated in place.  Requirements: n <= 256 (16 envs per workgroup, <= 16 actor workgroups + 1 on one XCD).
 * Layout of the layers in `params` (float offsets): w0 [128][4], b0 [128]; w1 [256][128] / b1 [256] = the stacked branch
 * layer (rows 0..127 actor, 128..255 critic); wa [2][128], ba [2]; wc [1][128], bc [1].
 * Replaces, for this class, T calls of xrl_rollout_step_cartpole + the bootstrap pass (same numbers up to fp32 summation
 * order: 16x16x4 tiles, per-workgroup partial sums of the observation statistics). */
typedef struct {
    const float* params;
    int32_t w0, b0, w1, b1, wa, ba, wc, bc;
    int32_t act;                                       /* hidden activation (XRL_ACT_*) */
    int32_t n, T, t0, n_steps;
    int32_t max_steps, use_obsnorm, use_rewnorm;
    int32_t flags;                                     /* bit 0: exchange through device-scope stores whatever the placement (tests) */
    float obs_range, rew_range, gamma, pad0;
    uint64_t seed, env_seed;
    uint32_t step; uint32_t pad1;                      /* Philox step of vector step t: step + *step_dev + t */
    const uint32_t* step_dev;
    /* state, updated in place */
    float* obs_raw;                                    /* [n][4] raw observations the agent acts on next */
    float* obs_stats; double* obs_count;               /* [8] mean | var, [1] */
    float* ret_stats; double* ret_count;               /* [2] mean, var, [1] */
    float* ret_track;                                  /* [n] discounted return tracker */
    double* cp_state; int32_t* cp_steps; int32_t* cp_episodes; float* cp_score; double* cp_stats;
    /* rollout-buffer fields [T][n] ([T][n][4] observations) */
    float* f_obs; float* f_act; float* f_logp; float* f_rew; float* f_term; uint8_t* f_seg; float* f_val; float* bootv;
    /* scratch */
    float* xnext;                                      /* [T][n][4] normalised next observations (before an auto-reset) */
    uint8_t* ended;                                    /* [T][n4] episode ended at this step (n4 = n rounded up to 4; zero-initialised) */
    float* ret_final;                                  /* [T][n4] discounted return of an episode that ended at this step */
    uint32_t* xchg;                                    /* [2048] exchange words (zeroed by the call, on the stream) */
    int32_t* status;                                   /* [4] zero-initialised by the caller: [0] != 0 -> a wait timed out, results invalid;
                                                        * [1] XCC id of workgroup 0, [2] mask of every XCC id ever seen, [3] launches that
                                                        * exchanged through device-scope stores */
    long long* dbg;                                    /* NULL, or [16]: shader-clock stamps of workgroup 0's chain wave at step n_steps / 2 */
    /* Tape provider (round 6; parity replays of recorded runs, envs/recorded.py: TapeCartPoleVecEnv).  tape_next_obs != NULL: what
     * envs.step() returned comes from these arrays instead of the in-kernel CartPole physics / reset draws -- row *tape_pos + t of
     * the tape for vector step t of the rollout: next observation (before an auto-reset), terminated / truncated flags, the first
     * observation of the next episode where either is set (infos[i]["reset_obs"], dummy_vec_env.py:65-76).  The reward stays the
     * class's constant 1 (CartPole-v1).  tape_u != NULL: the sampling uniform of (step t, env e) is tape_u[t * n + e] instead of
     * the Philox draw (Categorical.sample by inverse CDF).  Everything else -- statistics, normalisation, the actor, log-probs,
     * path ends, return tracker, reward normalisation, records -- is the same instruction stream as without a tape. */
    const float* tape_next_obs;                        /* [tape_rows][n][4] */
    const float* tape_reset_obs;                       /* [tape_rows][n][4] */
    const float* tape_term; const float* tape_trunc;   /* [tape_rows][n], 0 / 1 */
    const uint32_t* tape_pos;                          /* [1] tape row of the rollout's vector step 0 */
    const float* tape_u;                               /* NULL or [T][n] */
    int32_t tape_rows, pad2;
} xrl_rollout_run_t;
int xrl_rollout_cartpole_run(const xrl_rollout_run_t* p, xrl_stream_t stream);
/* Largest n_envs the whole-rollout launches accept on THIS device: their workgroups (16 envs each + one bookkeeper) must all be
 * resident on one XCD, one per CU (cu_count / 8 - 1 actor workgroups; 256 envs on an MI355X, fewer on a partitioned or smaller
 * device).  Callers ask before choosing the one-launch rollout (ops.CartPoleRollout.eligible / WideRollout.eligible) and fall back
 * to the launches per vector step above it; xrl_rollout_*_run itself answers XRL_EINVAL. */
int xrl_rollout_cartpole_max_envs(void);
int xrl_rollout_wide_max_envs(void);
int xrl_rollout_cartpole_values(const xrl_rollout_run_t* p, xrl_stream_t stream);
/* ------------------------------------------------------------------ rollout of the two-branch Gaussian class (csrc/rollout_wide.hip)
 * Vector steps [t0, t0 + n_steps) of a rollout of horizon T of PPO's step loop (ppo_agent.py:111-177) for the network class
 * D-256-256-{A | 1} (configs/ppo/mujoco.yaml) on the device-resident continuous-control provider (the dynamics of
 * xrl_synth_control_step), ONE launch, only the actor on the step chain: obs_rms.update + normalisation, actor forward (fp32 MFMA,
 * 16-row tiles, the 256 x 256 layer in registers), Normal(mu, std).sample() + log-prob (Philox streams of xrl_policy_sample),
 * dynamics with auto-reset, path-end flags, return tracker, normalised rewards + ret_rms.update in env order (a trailing
 * workgroup).  Values / bootstrap values are the caller's batched pass over f_obs / xnext afterwards.  State in place; n <= 256. */
typedef struct {
    const float* params;
    int32_t w0, b0, w1, b1, w2, b2, log_std_off;       /* actor branch: [256][D], [256], [256][256], [256], [A][256], [A]; log_std [A] */
    int32_t act, out_act;                              /* hidden activation; activation_action on the mean (none | tanh) */
    int32_t D, A, H;                                   /* D <= 20, A <= 8, H == 256 */
    int32_t n, T, t0, n_steps;
    int32_t max_steps, use_obsnorm, use_rewnorm, flags;
    float obs_range, rew_range, gamma, pad0;
    uint64_t seed, env_seed;
    uint32_t step, env_step;                           /* Philox step of vector step t: step + *step_dev + t (policy), env_step + *env_step_dev + t */
    const uint32_t* step_dev; const uint32_t* env_step_dev;
    /* state, updated in place */
    float* obs_raw;                                    /* [n][D] raw observations the agent acts on next */
    float* obs_mean; float* obs_var; double* obs_count;   /* [D], [D], [1] */
    float* ret_mean; float* ret_var; double* ret_count;   /* [1] each */
    float* ret_track;                                  /* [n] */
    float* env_state; int32_t* env_steps; float* env_score; double* env_stats;   /* [n][D], [n], [n], [4] */
    const float* Amat; const float* Bmat;              /* [D][D], [A][D] */
    /* rollout-buffer fields [T][n] (observations [T][n][D], actions [T][n][A]) */
    float* f_obs; float* f_act; float* f_logp; float* f_rew; float* f_term; uint8_t* f_seg;
    /* scratch */
    float* xnext;                                      /* [T][n][D] normalised next observations (before an auto-reset) */
    uint8_t* ended; float* ret_final; float* raw_rew;  /* [T][n4] each (n4 = n rounded up to 4; `ended` zero-initialised) */
    uint32_t* xchg;                                    /* [xrl_rollout_wide_words()] exchange words (zeroed by the call) */
    int32_t* status;                                   /* [4] as xrl_rollout_run_t.status */
    long long* dbg;                                    /* NULL, or [16] shader-clock stamps of workgroup 0 at step n_steps / 2 */
    /* Tape provider (round 6; envs/recorded.py: TapeControlVecEnv), as xrl_rollout_run_t.tape_*: tape_next_obs != NULL -> row
     * *tape_pos + t of the tape is what envs.step() returned at vector step t (next observation before an auto-reset, reward,
     * terminated / truncated, infos[i]["reset_obs"]) instead of the in-kernel provider dynamics; a terminated path then closes with
     * value 0 (seg bit pattern of xrl_rollout_cartpole_run).  tape_z != NULL: the standard normal of (step t, env e, action j) is
     * tape_z[(t * n + e) * A + j] instead of the Philox draw (action = mean + std * z, distributions.py:165-178). */
    const float* tape_next_obs;                        /* [tape_rows][n][D] */
    const float* tape_reset_obs;                       /* [tape_rows][n][D] */
    const float* tape_rew; const float* tape_term; const float* tape_trunc;   /* [tape_rows][n] */
    const uint32_t* tape_pos;                          /* [1] tape row of the rollout's vector step 0 */
    const float* tape_z;                               /* NULL or [T][n][A] */
    int32_t tape_rows, pad2;
} xrl_rollout_wide_t;
int xrl_rollout_wide_run(const xrl_rollout_wide_t* p, xrl_stream_t stream);
int xrl_rollout_wide_words(void);
/* dst[i] = src[i * ld] for i < n (a column of a row-major matrix: the value column of the batched head outputs -> buffer field). */
int xrl_copy_column(const float* src, int ld, float* dst, int64_t n, xrl_stream_t stream);
/* The shape-specialised kernel families (the rollout above, the shared-trunk minibatch kernels) are selected automatically
 * when the network is of their class; 0 forces the any-shape kernels (parity tests). */
int xrl_set_fast_kernels(int enable);
/* Re-pack the small parameters (first layer, biases, merged heads) into the image the step kernel copies to LDS with
 * one round trip; call once per rollout after the parameters changed.  Only params/layers/levels of *p are read. */
int xrl_pack_rollout_cache(const xrl_rollout_step_t* p, float* image, int64_t image_floats, xrl_stream_t stream);
/* Same, plus `frag` (N*K floats of the first big middle layer): its weights re-ordered so that wave w / k-chunk q / lane l
 * reads 16 contiguous bytes at ((w*K/8 + q)*64 + l)*16 -- every prefetch instruction of the step kernel is one 1 KB line run. */
int xrl_pack_rollout_cache2(const xrl_rollout_step_t* p, float* image, int64_t image_floats, float* frag, xrl_stream_t stream);
int64_t xrl_rollout_cache_floats(const xrl_rollout_step_t* p);

/* ------------------------------------------------------------------ fused PPO minibatch (ONE launch: gather -> MLP forward
 * -> PPO-clip loss -> MLP backward incl. weight gradients) for an actor-critic MLP with a categorical head.
 * A workgroup owns a 32-row tile of the minibatch and keeps every activation and gradient level in LDS; the weight
 * gradients of the middle layers are accumulated in MFMA registers and written once as this workgroup's slab.
 * Replaces, per minibatch, memory.sample (memory_tools.py:267-287) + PPO_Learner.update's forward/loss/backward
 * (ppo_learner.py:46-62); xrl_grad_reduce + xrl_adam_step finish the step.  Same numbers as the layered path
 * xrl_soa_gather -> xrl_linear_fwd x L -> xrl_ppo_loss_categorical -> xrl_linear_bwd_* (fp32 rounding order differs). */
typedef struct {
    const float* params;        /* flat parameters */
    const float* params_t;      /* same layout, middle-layer weights stored transposed ([K][N]) -- xrl_transpose_mid */
    const float* cache_image;   /* xrl_pack_rollout_cache image (first layer, biases, merged heads) */
    xrl_fused_layer_t layers[XRL_FUSED_MAX_LAYERS];
    int32_t n_layers, n_levels, n_head_layers;
    int32_t pad0;               /* tile_rows: 0 / 32 = one gradient slab per 32-row tile; 64 (with l0_fold_off, the role-split layout) =
                                 * 64-row tiles in the shared-trunk kernel (ppo_trunk.hip): slab and partials rows per 64 rows --
                                 * ceil(M/64) slabs, partials [2 ceil(M/64)][8] */
    int32_t level_width[XRL_FUSED_MAX_LEVELS];
    /* rollout buffer fields [T][n_envs][...] and the minibatch's env-major flat indices (memory_tools.py:270) */
    const float* f_obs; const float* f_act; const float* f_ret; const float* f_adv; const float* f_logp;
    const int64_t* idx;
    const float* stats;         /* NULL or (mean, std) of this minibatch's advantages: normalised on the fly */
    float* slabs;               /* [n_tiles][slab_stride] per-workgroup gradient partials, layout of params */
    double* partials;           /* [n_tiles][8] like xrl_ppo_loss_t.partials */
    float* diag;                /* NULL or [4][M] */
    int64_t slab_stride;
    int32_t M, n_envs, T, D, A;
    int32_t l0_fold_off;        /* 0, or the slab column (>= P, multiple of 4) of a (128 D + 128)-float fold region: selects the
                                 * role-split shared-trunk kernel (ppo_trunk.hip: TWO workgroups per tile, partials [2 n_tiles][8]; the
                                 * critic role's first-layer gradient goes to the fold region -- reduce with xrl_mirrors_t.fold_off / _len) */
    float clip_range, vf_coef, ent_coef, pad2;
    long long* dbg;             /* NULL, or [16] shader-clock stamps of the last workgroup (diagnostics) */
    const float* frag_image;    /* NULL, or xrl_pack_mid_frags copy of the first middle layer in MFMA B-fragment order */
    const float* f_rows;        /* NULL, or this minibatch's records already gathered, [M][8] (xrl_gather_rows): the kernel's
                                 * first load then needs neither the index nor a dependent second hop */
    const float* f_packed;      /* NULL, or xrl_pack_transitions records [T*n_envs][8] = obs[4] | act | ret | adv | old_logp:
                                 * one 32-byte random access per sampled row instead of five (D == 4 only) */
    /* the shared-trunk family D-128-{128-A | 128-1} (csrc/ppo_trunk.hip; selected with l0_fold_off > 0): */
    int32_t dist;               /* 0 categorical (f_act: [T][n_envs] action indices as float), 1 Gaussian (f_act: [T][n_envs][A]) */
    int32_t out_act;            /* Gaussian: activation_action on the mean, XRL_ACT_NONE | XRL_ACT_TANH (actor_head.py:62) */
    int32_t log_std_off;        /* Gaussian: float offset of actor.log_std [A] in params / a slab row */
    int32_t pad3;
    const uint16_t* frag16;     /* NULL, or the xrl_pack_mid_frags16 image of the branch layer (three bf16 planes): with 64-row tiles and
                                 * D <= 8, A <= 4 (CartPole (4, 2) -- the headline --, Acrobot, LunarLander, MountainCar; Gaussian: Pendulum) the
                                 * minibatch launch then forms its three 128-wide products as exact 3-way bf16 splits on the matrix cores
                                 * (csrc/ppo_trunk_bx.hip) */
    float* fwd_out;             /* xrl_trunk_forward16 only: [M][fwd_ld] head buffer -- columns [0, A) the actor's output, column A the value */
    int32_t fwd_ld, pad4;
} xrl_ppo_fused_t;
int xrl_ppo_fused_minibatch(const xrl_ppo_fused_t* p, xrl_stream_t stream);
/* The minibatch launch of the shared-trunk family (l0_fold_off > 0) CHAINED to the optimiser step of the minibatch before it:
 * the launch's workgroups first do what xrl_reduce_adam would have done for the PREVIOUS minibatch's slabs -- slab sums in
 * xrl_reduce_adam's order, global-norm clip, Adam, mirror maps: the same statements on the same elements, bit-identical
 * parameters / moments / state -- meet at two in-launch barriers (the norm; the new parameters, agent-scope release / acquire)
 * and then run the minibatch on the updated parameters.  Per minibatch this replaces the pair {xrl_ppo_fused_minibatch,
 * xrl_reduce_adam} (train_epochs' loss.backward(); clip_grad_norm_; optimizer.step(); scheduler.step() -- ppo_learner.py:46-67)
 * by ONE launch: one kernel boundary, one argument fetch and one cold start fewer, and the optimiser's blocks no longer wait for
 * a launch of their own.  An update phase is: xrl_ppo_fused_minibatch (first minibatch), xrl_ppo_trunk_chained x (n - 1),
 * xrl_reduce_adam (the last minibatch's step).
 * opt->slabs must hold the previous launch's n_split slab rows (normally the same buffer as p->slabs: every workgroup is past
 * the first barrier -- and with it past its slab reads -- before any workgroup writes a new slab).  Needs every workgroup of
 * the launch resident at once (xrl_ppo_trunk_chain_fits), 2 * ceil(M / tile_rows) >= ceil(P / 256) workgroups, P % 4 == 0.
 * Of opt->mirrors only map / dst / n and fold_off / fold_len are honoured (anything else set: XRL_EINVAL).
 * opt->sync: [XRL_CHAIN_SYNC_WORDS] uint32, zero-initialised once; sync[2] != 0 afterwards = a barrier timed out (the phase's
 * updates are invalid from that launch on; the state's norm reads NaN). */
#define XRL_CHAIN_MAX_WGS 512
#define XRL_CHAIN_SYNC_WORDS (4 + 2 * XRL_CHAIN_MAX_WGS)
typedef struct {
    const float* slabs;          /* the previous minibatch's gradient slabs [n_split][slab_stride] */
    int64_t slab_stride;
    float* params; float* grad; float* m; float* v;
    int64_t P;
    xrl_adam_state_t* state;
    double* sumsq_part;          /* [n_part] */
    double max_norm;             /* <= 0: no clipping (the barriers stay: every workgroup needs the new parameters) */
    uint32_t* sync;
    int32_t n_split, n_part;
    xrl_mirrors_t mirrors;
} xrl_opt_chain_t;
int xrl_ppo_trunk_chained(const xrl_ppo_fused_t* p, const xrl_opt_chain_t* opt, xrl_stream_t stream);
/* 1 if xrl_ppo_trunk_chained can run a minibatch of M rows in tiles of tile_rows (32 | 64) for P parameters on this device */
int xrl_ppo_trunk_chain_fits(int32_t M, int32_t tile_rows, int64_t P);
/* packed[i][0..7] = obs[i][0..3], act[i], ret[i], adv[i], logp[i] for i < count (the rollout buffer's [t][env] order):
 * the record form the fused minibatch kernel gathers from; run once per update phase after the advantages exist. */
int xrl_pack_transitions(const float* f_obs, const float* f_act, const float* f_ret, const float* f_adv,
                         const float* f_logp, float* packed, int64_t count, xrl_stream_t stream);
/* out[i][0..7] = packed[(idx[i] % T) * n_envs + idx[i] / T][0..7] for i < count: every minibatch of an update phase gathered
 * in one launch (idx = all n_epochs x n_minibatch x batch indices, env-major flat indices as in memory_tools.py:270). */
int xrl_gather_rows(const float* packed, const int64_t* idx, float* out, int64_t count, int n_envs, int T, xrl_stream_t stream);
/* frag <- the first middle layer W[N][K] (N % 32 == 0, K % 32 == 0) in the order the matrix-core kernels consume it, so
 * that every prefetch instruction of a wave reads one contiguous 1 KB run:
 *   forward section  [N/32][K/8][64 lanes][4]: W[32 t + (l & 31)][8 q + 4 (l >> 5) + s] in slot (q + t) mod K/8 of tile t
 *   backward section [K/32][N/8][64 lanes][4]: W[8 q + 4 (l >> 5) + s][32 t + (l & 31)] in slot (q + 2t) mod N/8 (offset N*K)
 * (the slot rotation spreads the simultaneous fetches of a workgroup's waves over all L2 channels)
 * Only params/layers of *p are read; frag_floats >= 2*N*K. */
int xrl_pack_mid_frags(const xrl_ppo_fused_t* p, float* frag, int64_t frag_floats, xrl_stream_t stream);
/* image <- the stacked branch layer W[256][128] of the shared-trunk family as THREE bf16 planes h | m | l (XRL_FRAG16_PLANE elements
 * each) with  W[n][k] == h + m + l  exactly:  h = bf16(w), m = bf16(w - h), l = bf16(w - h - m), round-to-nearest-even.  Inside a
 * plane, in the order the lanes of v_mfma_f32_32x32x16_bf16 consume it (a lane's 8 elements contiguous, a wave's load 1 KB):
 *   forward section  [8 tiles t][8 k-steps][64 lanes][8]: W[32 t + (l & 31)][16 qq + 8 (l >> 5) + e] in slot (qq + t) mod 8 of tile t
 *   backward section [4 tiles kt][16 n-steps][64 lanes][8]: W[16 q + 8 (l >> 5) + e][32 kt + (l & 31)] in slot (q + kt) mod 16 (offset 256*128)
 * The optimiser launches keep it current through SPLIT mirror maps (xrl_mirrors_t.split_plane).  image_elems >= 3 * XRL_FRAG16_PLANE.
 * Replaces nothing in the reference: it is a derived layout of policy.actor / critic first Linear (policies/categorical.py ActorCriticPolicy). */
#define XRL_FRAG16_PLANE (2 * 256 * 128)
int xrl_pack_mid_frags16(const xrl_ppo_fused_t* p, uint16_t* image, int64_t image_elems, xrl_stream_t stream);
/* diagnostics: the split-product kernel's weight-gradient operands through ds_read_b64_tr_b16 (1, default) or 2-byte LDS reads (0);
 * same numbers either way. */
/* The ACTING pass of the shared-trunk networks with D <= 8, A <= 4 as ONE launch (replaces the three launches of the layered forward:
 * ActorCriticPolicy.forward of policies/categorical.py / gaussian.py -- representation, actor head, critic head; on_policy.py:128-169
 * calls it once per vector step): row m of p->f_obs [M][D] (contiguous) -> p->fwd_out[m][0..A) = logits | activation_action(mean),
 * p->fwd_out[m][A] = value.  Reads params, layers, frag16 (must be current: xrl_pack_mid_frags16 or the optimiser's split mirror stores),
 * M, D, A, dist, out_act, fwd_out, fwd_ld of *p; the 128-wide product runs as the exact 3-way bf16 split of the minibatch kernel.
 * sample != NULL (sample->n rows of observations, M == n or 2 n; .heads / .ld ignored): the launch also does xrl_policy_sample's work
 * (get_actions, on_policy.py:128-169; the same statements, csrc/sample.h) and writes NO head buffer -- the actor role samples
 * act_out / env_action / logp_out of rows [0, n), the critic role writes val_out of rows [0, n) and bootv_prev from rows [n, 2 n);
 * act_out == NULL: bootstrap values only. */
int xrl_trunk_forward16(const xrl_ppo_fused_t* p, const xrl_sample_t* sample, xrl_stream_t stream);
/* The whole-rollout / per-step actor kernel of the CartPole class (xrl_rollout_cartpole_run): its 128-deep branch-layer products as exact
 * 3-way bf16 splits on v_mfma_f32_16x16x32_bf16 (1) or on the float32 instruction (0); see csrc/rollout_actor.hip. */
int xrl_set_rollout_split_products(int on);
int xrl_set_split_product_tr(int on);
/* diagnostics: which weight-streamed products of the split-product kernel have the two waves of a 32-column block split the k-range
 * (each streams half of the fragment planes, the halves meet through LDS) instead of the rows: 0 none, 1 the backward-data product
 * (default), 2 both.  Another summation order: results differ by float32 rounding (and, with 2, by what that does to act'(h2) of
 * pre-activations within rounding noise of zero: csrc/ppo_trunk_bx.hip). */
int xrl_set_split_product_ksplit(int mode);
/* ------------------------------------------------------------------ fused PPO minibatch, two-branch Gaussian actor-critic
 * D-256-256-{A | 1} (configs/ppo/mujoco.yaml:8-13: Basic_Identical representation, actor / critic hidden [256, 256];
 * policies/gaussian.py ActorCriticPolicy, ppo_learner.py:46-62).  ONE launch per minibatch, two workgroups (actor branch,
 * critic branch) per 32-row tile; rows are read from staged arrays (the caller gathered the minibatch, e.g. every minibatch of
 * an update phase with one xrl_soa_gather).  Gradients go to slab row `tile` in the layout of params (both roles write
 * disjoint ranges); reduce with xrl_reduce_adam(n_split = ceil(M/32)).  partials: [2*ceil(M/32)][8], row 2*tile + role. */
typedef struct {
    int32_t w0, b0, w1, b1, w2, b2;                    /* float offsets of the branch's three layers in params / a slab row */
} xrl_wide_branch_t;
typedef struct {
    const float* params;
    const float* frag;                                 /* xrl_ppo_wide_pack image: [2][2][256*256] middle layers in B-fragment order */
    xrl_wide_branch_t br[2];                           /* 0: actor (D-256-256-A), 1: critic (D-256-256-1) */
    int32_t log_std_off;
    int32_t D, A, H;                                   /* D <= 24, A <= 8, H == 256 */
    int32_t act, out_act;                              /* hidden activation (relu | leaky_relu | tanh), activation_action (none | tanh) */
    int32_t M, dbg_role;
    const float* obs;                                  /* [M][D], 16-byte aligned */
    const float* actions;                              /* [M][A], 16-byte aligned */
    const float* ret; const float* adv; const float* old_logp;   /* [M] */
    const float* stats;                                /* NULL or (mean, std) of this minibatch's advantages */
    float* slabs; int64_t slab_stride;
    double* partials;
    float* diag;                                       /* NULL or [4][M]: log_prob, ratio, surrogate1, surrogate2 */
    float* heads;                                      /* NULL or [M][A + 1]: actor output (after activation_action) | value --
                                                        * the callback tensors a_dist / v_pred of ppo_learner.py:82-88 */
    float clip_range, vf_coef, ent_coef, pad0;
    long long* dbg;                                    /* NULL, or [16] shader-clock stamps of the last tile's role dbg_role */
    /* NULL, or [2][rows_ld][256] each: the launch then leaves the middle layer's weight gradient to xrl_wide_dw1 -- every workgroup
     * stores its tile's 32 rows of dLoss/d(pre-activation of h2) and of h1 here (64 KB) instead of forming and storing a 256 x 256
     * partial of dW1 (256 KB per workgroup, 92 % of the 73 MB of gradient rows a 4 096-row minibatch wrote), and writes nothing
     * to the w1 ranges of its slab row. */
    float* rows_g2; float* rows_h1;
    int64_t rows_ld;                                   /* rows per branch in rows_g2 / rows_h1, >= 32 * ceil(M / 32) */
} xrl_ppo_wide_t;
int xrl_ppo_wide_minibatch(const xrl_ppo_wide_t* p, xrl_stream_t stream);
/* dW1 of both branches from the rows xrl_ppo_wide_minibatch left (rows_g2, rows_h1): split over the rows in ceil(32 ceil(M / 32) /
 * 128) parts of 128 rows; part s goes to the w1 ranges of slab row s (dW1[n][k] = sum over the part's rows of g2[row][n] *
 * h1[row][k], rows in order).  xrl_reduce_adam then sums that many rows for those ranges (xrl_mirrors_t.alt_*).  Returns the
 * number of parts through *n_parts. */
int xrl_wide_dw1(const xrl_ppo_wide_t* p, int32_t* n_parts, xrl_stream_t stream);
/* frag[b][0][t][(q + t) mod 32][l][s] = W1_b[32 t + (l & 31)][8 q + 4 (l >> 5) + s]   forward section
 * frag[b][1][t][(q + t) mod 32][l][s] = W1_b[8 q + 4 (l >> 5) + s][32 t + (l & 31)]   backward section (reduction index on k)
 * (b: branch, t: 32-wide output tile, q: 8-wide chunk of the reduction index, l: lane): every prefetch instruction of a wave
 * reads one contiguous 1 KB run.  Only params / br of *p are read; frag holds 4*256*256 floats.  Kept current afterwards by
 * the optimiser launch (two xrl_mirrors_t maps built from this layout: one per section). */
int xrl_ppo_wide_pack(const xrl_ppo_wide_t* p, float* frag, xrl_stream_t stream);
/* Acting step of the same network class in ONE launch (replaces three grouped GEMM launches + xrl_policy_sample of the layered
 * rollout step; on_policy agent: policy(obs) -> stochastic_sample / log_prob / values, ppo_agent.py:97-135): x holds 2n
 * normalised observations, rows [0, n) the ones acted on, rows [n, 2n) the previous step's next observations whose values
 * bootstrap truncated episodes.  flags bit 0: act -- action = mu + std * N(0,1) (the Philox draws of xrl_policy_sample: key
 * seed, counter (row, step + *step_dev, STREAM_GAUSS + dim)), log-prob and value of rows [0, n); bit 1: values of rows
 * [n, 2n) -> bootv_prev.  Same numbers as the layered sequence up to the summation order inside a dot product.  Grid: four
 * workgroups per (32-row tile, branch), one per 64 output columns of the middle layer (a CU streams only ~25 GB/s of
 * weights); the last of the four to finish adds the head's partial sums in part order and samples. */
typedef struct {
    const float* params;
    const float* frag;                                 /* xrl_ppo_wide_pack image */
    xrl_wide_branch_t br[2];
    int32_t log_std_off;
    int32_t D, A, H;
    int32_t act, out_act;
    int32_t n, flags;
    const float* x;                                    /* [2n][D] */
    float* act_out;                                    /* [n][A] */
    float* env_action_f;                               /* NULL or [n][A]: the same actions for the vector env */
    float* logp_out;                                   /* [n] */
    float* val_out;                                    /* NULL or [n] */
    float* bootv_prev;                                 /* [n] (flags bit 1) */
    uint64_t seed;
    uint32_t step, pad0;
    const uint32_t* step_dev;
    /* Optional: rows [0, n) from RAW observations -- RunningMeanStd.update + _process_observation inside this launch
     * (what xrl_obs_normalize does, the same arithmetic; statistic_tools.py:117-185, agent.py:262-283).  raw != NULL: rows
     * [0, n) of x are ignored; the statistics are read from *_in and written to *_out (different buffers: the workgroups
     * of a launch read the old ones while one of them stores the new ones -- the caller alternates two sets);
     * n <= 4 * floor(1024 / D); n % 32 == 0 when rows [n, 2n) are evaluated in the same launch. */
    const float* raw;                                  /* NULL or [n][D] */
    const float* mean_in; const float* var_in; const double* count_in;
    float* mean_out; float* var_out; double* count_out;
    float* obs_slot;                                   /* NULL or [n][D]: the normalised rows (the rollout buffer's slot) */
    int32_t update, normalize;                         /* as xrl_rms_t */
    float range, pad1;
    /* scratch the four workgroups of a (tile, branch) pair meet in: xchg [pairs][4][32][8] floats, xcnt [pairs] uint32
     * zero-initialised once (self-resetting); pairs = ceil(n/32) actor + up to ceil(2n/32) critic tiles */
    float* xchg; uint32_t* xcnt;
    /* Optional: rows [n, 2n) from the RAW next observations of the previous vector step (NULL: from x), normalised with
     * (mean_in, var_in) when `normalize` -- and that step's bookkeeping (xrl_rollout_poststep with next_obs_norm = NULL) done
     * by one extra workgroup of this launch.  n % 32 == 0. */
    const float* next_raw;
    xrl_poststep_t post;
    int32_t has_post, pad2;
    long long* dbg;                                    /* NULL, or [32] shader-clock stamps of the four workgroups of pair 0 (diagnostics) */
} xrl_wide_act_t;
int xrl_wide_act_step(const xrl_wide_act_t* p, xrl_stream_t stream);
/* params_t <- params with every middle layer's weight transposed (call after each optimiser step). */
int xrl_transpose_mid(const xrl_ppo_fused_t* p, float* params_t, xrl_stream_t stream);

/* ------------------------------------------------------------------ TD targets: DQN and QMIX */

/* DQN_Learner.update loss head (qlearning_family/dqn_learner.py:39-46): predictQ = gather(evalQ, a);
 * y = r + gamma (1-d) max_a' targetQ  (or targetQ[argmax_a' q_next_eval] when q_next_eval != NULL, the DDQN rule of
 * ddqn_learner.py:39-47); loss = mean((predictQ - y)^2); d_q = dLoss/d evalQ. */
typedef struct {
    const float* q_eval;       /* [M][ld] Q(s,.) of the eval network */
    const float* q_next;       /* [M][ld] Q_target(s',.) */
    const float* q_next_eval;  /* NULL (DQN) or [M][ld] Q_eval(s',.) (double-Q action selection) */
    const float* actions;      /* [M] f32 action indices (the replay buffer stores actions as float32) */
    const float* rewards;      /* [M] */
    const float* terminals;    /* [M] f32 0/1 */
    float* d_q;                /* [M][ld] */
    float* diag;               /* NULL or [2][M]: predictQ, targetQ (callback tensors dqn_learner.py:72-74) */
    double* partials;          /* [n_split][8]: sum (predictQ-y)^2, sum predictQ, 0... */
    int32_t M, A, ld, n_split;
    float gamma;
    int32_t dueling;           /* != 0: head rows are [advantages (A) | value] and Q = V + (A_j - mean A)
                                * (DuelingQValueHead, rl_models/heads/q_head.py:42-80; dueldqn_learner.py:28-75); ld >= A + 1 */
    float huber_delta;         /* <= 0: loss = mean((predictQ - y)^2), the reference's DQN default (dqn_learner.py:25,46);
                                * > 0: loss = nn.HuberLoss(delta = huber_delta, reduction "mean")(predictQ, y), the form of the
                                * reference's `use_huber_loss` / `huber_delta` switch (learners/base/marl_learner.py:193-197);
                                * partials[.][0] then sums the Huber terms.  Same field in xrl_dqn_head_td_t / xrl_dqn_tail_td_t. */
    int32_t pad;
} xrl_dqn_td_t;
int xrl_dqn_td(const xrl_dqn_td_t* p, xrl_stream_t stream);
/* The last layer of a BasicQhead (q_head.py:8-39: Linear(H, n_actions)), xrl_dqn_td's rule and the layer's data gradient in ONE
 * launch: q = h . W^T + b for the eval network on obs (rows [0, M) of h_eval; rows [M, 2M) = next_obs under double-Q) and the target
 * network on next_obs, TD target / loss / dQ as xrl_dqn_td, d_h[m] = dQ[m][a_m] * W_eval[a_m] * act'(h_eval[m]).  partials: [M][8],
 * one row per transition ({td^2, predictQ, 0...}). */
typedef struct {
    const float* h_eval;       /* [M or 2M][ld_h] hidden activations in front of the Q layer (outputs of `act`) */
    const float* h_target;     /* [M][ld_h] */
    const float* w_eval;       /* [A][H] */
    const float* b_eval;       /* [A] */
    const float* w_target;
    const float* b_target;
    const float* actions;      /* [M] f32 */
    const float* rewards;
    const float* terminals;
    float* q_eval;             /* [M or 2M][ld_q] written */
    float* q_target;           /* [M][ld_q] written */
    float* d_q;                /* [M][ld_q] */
    float* d_h;                /* [M][ld_h] */
    float* diag;               /* NULL or [2M]: predictQ | targetQ */
    double* partials;          /* [M][8] */
    int32_t M, A, H, ld_h, ld_q, double_q, act, pad;
    float gamma, huber_delta;  /* huber_delta: see xrl_dqn_td_t */
} xrl_dqn_head_td_t;

/* xrl_dqn_head_td extended backwards to the last convolution's output and forwards to the convolution stack's incoming gradient
 * (Basic_CNN, cnn.py:11-50: AdaptiveMaxPool2d((1,1)) over P = OH*OW positions of F = 64 filters; BasicQhead F -> H -> n_actions,
 * q_head.py:8-39): pool of the frames of transition m, hidden layer of the eval and the target network, xrl_dqn_head_td's part,
 * d_feat = d_h . W1, and the pool's backward into dY of the last convolution -- one launch, one workgroup per transition. */
typedef struct {
    const float* y_eval;       /* [M or 2M frames][P][F] last convolution's (relu) output, eval network: obs | next_obs (double-Q) */
    const float* y_target;     /* [M][P][F] target network on next_obs */
    float* feat_eval;          /* [M or 2M][ld_f] pooled features, written (the hidden layer's weight gradient reads rows [0, M)) */
    float* feat_target;        /* NULL or [M][ld_f] */
    int32_t* arg;              /* NULL or [M][F] first-maximum positions of the eval(obs) frames */
    const float* w1_eval; const float* b1_eval; const float* w1_target; const float* b1_target;   /* [H][F], [H] */
    const float* w2_eval; const float* b2_eval; const float* w2_target; const float* b2_target;   /* [A][H], [A] */
    const float* actions; const float* rewards; const float* terminals;                           /* [M] f32 */
    float* q_eval;             /* [M or 2M][ld_q] written */
    float* q_target;           /* [M][ld_q] written */
    float* d_q;                /* [M][ld_q] */
    float* h_eval;             /* [M or 2M][ld_h] hidden activations, written */
    float* d_h;                /* [M][ld_h] */
    float* d_feat;             /* NULL or [M][ld_f] */
    float* dy;                 /* [M][P][F] gradient w.r.t. the last convolution's pre-activation output */
    float* diag;               /* NULL or [2M] */
    double* partials;          /* [M][8] */
    int32_t M, A, H, F, P, ld_h, ld_q, ld_f, double_q, act;
    float gamma, huber_delta;  /* huber_delta: see xrl_dqn_td_t */
    /* NULL, or the gradient slabs of the optimiser launch, [>= M][slab_stride]: transition m writes ITS term of the two dense layers'
     * weight / bias gradients (d_h[m] x feat[m], d_h[m], d_q[m] x h[m], d_q[m]) into slab m at the parameters' offsets -- the
     * slab reduction of xrl_reduce_adam is then the sum over the batch (in order of m), and no weight-gradient GEMM is launched. */
    float* slabs;
    int64_t slab_stride, off_w1, off_b1, off_w2, off_b2;
} xrl_dqn_tail_td_t;
int xrl_dqn_tail_td(const xrl_dqn_tail_td_t* p, xrl_stream_t stream);

/* The acting side of the same network (OffPolicyAgent.exploration on DeepQNetwork.forward, off_policy.py:129-148,
 * deep_q_network.py:61-80): pool of env e's frame, hidden layer, Q layer, greedy action + epsilon coin as xrl_egreedy -- one launch. */
typedef struct {
    const float* y;            /* [n][P][F] last convolution's output */
    const float* w1; const float* b1; const float* w2; const float* b2;   /* [H][F], [H], [A][H], [A] */
    const float* eps_dev;      /* [1], or NULL: `eps` below */
    int32_t* action;           /* [n] */
    float* action_f;           /* NULL or [n] */
    float* q;                  /* NULL or [n][ld_q] */
    float* feat;               /* NULL or [n][ld_f] */
    const uint32_t* step_dev;
    uint64_t seed;
    uint32_t step;
    int32_t n, A, H, F, P, ld_q, ld_f, act;
    float eps;                 /* used when eps_dev is NULL and eps_sched is 0 */
    /* eps_sched != 0: epsilon of vector step k = step + *step_dev computed HERE, the host's own arithmetic (off_policy.py:119-127:
     * e_greedy = start_greedy - current_step * delta_egreedy in float64 while the previous value is above end_greedy; current_step
     * = k * eps_n): e = (float)(eps_start - (double)(min(k, eps_kstar) * eps_n) * eps_delta), eps_kstar = the first k whose value
     * is <= end_greedy (a host constant).  What a captured vector step needs: no argument changes from step to step. */
    int32_t eps_sched, eps_n;
    uint32_t eps_kstar, pad1;
    double eps_start, eps_delta;
} xrl_dqn_act_tail_t;
int xrl_dqn_act_tail(const xrl_dqn_act_tail_t* p, xrl_stream_t stream);
int xrl_dqn_head_td(const xrl_dqn_head_td_t* p, xrl_stream_t stream);

/* QMIX_Learner.update between the per-agent Q-networks and the hyper-networks
 * (multi_agent_rl/qmix_learner.py:34-86, iql_learner.py:63-81, heads/q_mix_head.py:66-95): gather taken Q, masked
 * (double-)Q target action, agent masks, monotonic mixing (abs / bmm / ELU) for the eval and target mixers, TD target,
 * MSE loss and the backward pass down to d Q_eval and d(hyper-network outputs).  One wavefront per batch row. */
typedef struct {
    const float* q_eval;       /* [B*N][ldq] Q_eval(obs), row b*N+n */
    const float* q_next_eval;  /* NULL or [B*N][ldq] Q_eval(next_obs): double-Q argmax (config.double_q) */
    const float* q_next;       /* [B*N][ldq] Q_target(next_obs) */
    const float* actions;      /* [B][N] f32 */
    const float* avail_next;   /* NULL or [B][N][A] f32 0/1 (use_actions_mask) */
    const float* agent_mask;   /* [B][N] f32 0/1 */
    const float* rewards;      /* [B][N] */
    const float* terminals;    /* [B][N] f32 0/1 */
    const float* e_b1;         /* [B][ld_e1] eval hyper_b_1(state) */
    const float* e_raw;        /* [B][ld_e2] eval raw hyper outputs: w1_raw[N*H] | w2_raw[H] | b2[1] */
    const float* t_b1;         /* target mixer, same layout, on state_next */
    const float* t_raw;
    float* d_q;                /* [B*N][ldq] dLoss/dQ_eval */
    float* d_e_b1;             /* [B][ld_e1] */
    float* d_e_raw;            /* [B][ld_e2] */
    float* diag;               /* NULL or [3][B]: q_tot_eval, q_tot_next, q_tot_target (qmix_learner.py:108-110) */
    double* partials;          /* [B][8]: (q_tot_eval-y)^2, q_tot_eval, 0... */
    int32_t B, N, A, H, ldq, ld_e1, ld_e2, ld_t1, ld_t2, double_q;
    float gamma;
    int32_t mixer;             /* 0: QMIX_Mixer (q_mix_head.py:28-95).  1: VDN_Mixer, the sum over agents (vdn_learner.py:13-106;
                                * e_, t_, d_e_ pointers unused).  2: IndependentMixer / IQL_Learner (iql_learner.py:85-142):
                                * per-agent TD (td * mask), loss sum(td^2)/sum(mask); partials[b] = {sum_n td^2, sum_n taken Q,
                                * sum_n mask}, diag [2][B*N] = taken Q, target (feed-forward only) */
    const float* filled;       /* NULL (feed-forward: MSE over B rows, qmix_learner.py:86) or [B] f32 0/1 step mask of the
                                * recurrent branch (:60-61 valid_mask = agent_mask * filled, :82-84 loss =
                                * sum((td * filled)^2) / sum(filled)); partials[b][2] = filled[b].  In that mode the B rows
                                * are the (t, episode) pairs of a time-major batch and the "next" arrays are the same
                                * arrays one time slot further (caller passes offset pointers). */
} xrl_qmix_t;
int xrl_qmix_mix_td(const xrl_qmix_t* p, xrl_stream_t stream);

/* The whole QMIX_Learner.update for feed-forward agents in ONE launch (qmix_learner.py:24-112 between the sampled batch and
 * optimizer.step(); heads/q_mix_head.py:28-95; value_factorization.py:66-150): per-agent Q networks eval(obs) /
 * eval(next_obs) / target(next_obs), masked (double-)Q target action, eval and target hyper-networks, monotonic mixing,
 * TD target, MSE loss, and the backward pass down to every weight gradient.  The batch is cut into groups of items_per_wg
 * transitions; workgroup g keeps its group's activations in LDS from the observations to the gradients and writes its
 * partial of every parameter gradient to slabs[g] (sum the ceil(B / items_per_wg) slabs with xrl_reduce_adam /
 * xrl_grad_reduce: fixed order, bit-deterministic).  Same numbers as the layered path (xrl_linear_fwd x3, xrl_qmix_mix_td,
 * xrl_linear_bwd_*) up to fp32 summation order.  Weights are read from two weight IMAGES (eval, target) that already have
 * the padded LDS layout, so staging them is a straight copy; the gradient of the parameter at flat offset o (w_off / b_off /
 * mix_off, torch nn.Linear layout [out][in]) is written to slab offset o. */
#define XRL_QF_MAX_LAYERS 4
enum { XRL_QF_FIRST_W = 0, XRL_QF_FIRST_B, XRL_QF_B1_W, XRL_QF_B1_B, XRL_QF_W1_W, XRL_QF_W1_B, XRL_QF_W2_W, XRL_QF_W2_B,
       XRL_QF_B2_W, XRL_QF_B2_B, XRL_QF_N_OFF };
typedef struct {          /* float offsets inside a weight image (xrl_qmix_fused_layout): agent block, then mixer block */
    int32_t w[XRL_QF_MAX_LAYERS], b[XRL_QF_MAX_LAYERS], ldw[XRL_QF_MAX_LAYERS];   /* agent layer l: W_l [N_l][ldw_l], b_l; block offsets */
    int32_t mw[5], mb[5], mldw[5];                 /* mixer: FIRST, B1, W1, W2, B2 (offsets inside the mixer block) */
    int32_t agent_floats, mixer_floats;            /* block sizes; the mixer block starts at agent_floats */
} xrl_qf_image_t;
typedef struct {
    const float* img_eval;      /* weight image of the eval networks  [agent_floats + mixer_floats], 16-byte aligned: element */
    const float* img_target;    /* (n, k) of a matrix at block offset w + n * ldw + k, padding zero; keep them current with
                                 * xrl_mirrors_t (map = parameter index -> image index) and .target_image */
    int32_t n_layers;           /* nn.Linear layers of the agent network (representation + Q head), <= XRL_QF_MAX_LAYERS */
    int32_t act;                /* XRL_ACT_* after every layer but the last */
    int32_t dims[XRL_QF_MAX_LAYERS + 1];   /* dims[0] = obs_dim ... dims[n_layers] = n_actions */
    int32_t products;           /* the matrix products: 0 = on the matrix cores (16x16x4 fp32 MFMA tiles) when a workgroup carries
                                 * items_per_wg * N >= 8 rows, else VALU loops; 1 = MFMA, 2 = VALU (parity tests) */
    int64_t w_off[XRL_QF_MAX_LAYERS], b_off[XRL_QF_MAX_LAYERS];
    int64_t mix_off[XRL_QF_N_OFF];          /* FIRST = [hyper_w_1.0; hyper_w_2.0; hyper_b_2.0] stacked ([3 HH][S], biases [3 HH]);
                                             * B1 = hyper_b_1 [H][S]; W1 = hyper_w_1.2 [N H][HH]; W2 = hyper_w_2.2 [H][HH];
                                             * B2 = hyper_b_2.2 [1][HH] */
    int32_t N, A, S, H, HH;     /* agents, actions, state dim, mixer embed dim (<= 64), hyper hidden dim */
    int32_t B, items_per_wg, double_q;
    const float* obs;           /* [B][N][obs_dim] */
    const float* obs_next;
    const float* state;         /* [B][S] */
    const float* state_next;
    const float* actions;       /* [B][N] f32 */
    const float* rewards;       /* [B][N] */
    const float* terminals;     /* [B][N] f32 0/1 */
    const float* agent_mask;    /* [B][N] f32 0/1 */
    const float* avail_next;    /* NULL or [B][N][A] f32 0/1 */
    float* slabs;               /* [ceil(B / items_per_wg)][slab_stride] */
    int64_t slab_stride;
    double* partials;           /* [B][8]: (q_tot_eval - y)^2, q_tot_eval, 0... (as xrl_qmix_mix_td) */
    float* diag;                /* NULL or [3][B]: q_tot_eval, q_tot_next, q_tot_target */
    float gamma, pad1;
    long long* dbg;             /* NULL, or [16] cycle-counter stamps of workgroup 0 at the phase boundaries (diagnostics) */
    /* ring mode (ring_n_envs > 0): the nine batch pointers above are the FIELDS of the replay ring, [n_size][n_envs][row]
     * (xrl_soa_store_step's layout), and transition b of the batch is the ring row xrl_sample_replay_indices' stream draws
     * for (seed, counter + *counter_dev, b): the launch samples and gathers for itself (memory_tools_marl.py:742-765) */
    int32_t ring_n_envs, ring_n_size;
    const int32_t* size_dev;        /* [1] filled ring slots */
    const uint32_t* counter_dev;    /* NULL or [1] */
    int64_t* idx_out;               /* NULL or [B]: the rows drawn (flat index env * n_size + step) */
    uint64_t draw_seed;
    uint32_t draw_counter, pad3;
} xrl_qmix_fused_t;
/* A whole update phase of the feed-forward QMIX learner -- train_epochs' n_epochs x {sample, update} (off_policy_marl.py:573-594 with
 * qmix_learner.py:24-112) -- as ONE launch (round 6): the workgroups of xrl_qmix_fused_update stay resident for the n_updates updates of
 * the phase; after every update they meet (flags through the L2 they share: the launch keeps blockIdx % 8 == 0 of an 8x
 * oversubscribed grid, i.e. one XCD's share, as the one-launch rollout kernels do; any other placement is noticed and goes through
 * agent-scope release / acquire fences), workgroup g sums its 1/n-th of the parameters over the slabs in xrl_reduce_adam's order,
 * applies Adam (+ LinearLR, + the periodic hard target update) and writes the new parameters to params, both weight images and the
 * acting launch's image; a second meeting, and every workgroup re-stages the weights for the next update.  Same statements per element
 * as n_updates x {xrl_qmix_fused_update(ring mode, counter + u), xrl_reduce_adam}: parameters, moments, target and images bit-identical.
 * Needs: ring mode (p->ring_n_envs > 0: update u draws with counter draw_counter + u), no gradient clipping (configs/qmix/sc2/3m.yaml:45:
 * nothing then depends on the global norm inside an update), ceil(B / items_per_wg) <= compute units / 8 workgroups, P % 4 == 0.
 * p->partials is ignored: update u writes its loss partials to phase_partials[u] and their sums to epoch_sums[u]. */
#define XRL_QF_PHASE_SYNC_WORDS 256
typedef struct {
    int32_t n_updates;
    int32_t sync_every;         /* > 0: target <- parameters whenever the optimiser step is a multiple of it (copy_target, qmix_learner.py:105-106) */
    float* params; float* grad; float* m; float* v;
    int64_t P;
    xrl_adam_state_t* state;
    const int32_t* map;         /* [P] parameter index -> index in the weight images (< 0: none); p->img_eval / p->img_target are written */
    float* target;              /* [P] target parameters */
    float* act_image;           /* NULL, or the acting launch's weight image ... */
    const int32_t* act_map;     /* ... and its map [P] */
    double* phase_partials;     /* [n_updates][B][8] */
    double* epoch_sums;         /* [n_updates][8] */
    double* sumsq_part;         /* [>= workgroups] partial sums of squares of the reduced gradient (last_grad_norm) */
    float* scalars;             /* [2 n_updates] scratch: Adam's step size / sqrt(bias correction 2) per update (filled by the call) */
    uint32_t* tick;             /* NULL, or a counter advanced by tick_inc at the end of the phase (the replay draw counter) */
    int32_t tick_inc, pad;
    uint32_t* sync;             /* [XRL_QF_PHASE_SYNC_WORDS] zero-initialised once; sync[2] != 0 afterwards: a wait timed out (phase invalid) */
} xrl_qmix_phase_t;
int xrl_qmix_fused_phase(const xrl_qmix_fused_t* p, const xrl_qmix_phase_t* phase, xrl_stream_t stream);
/* 1 if xrl_qmix_fused_phase can run B transitions in groups of items_per_wg for P parameters on this device */
int xrl_qmix_fused_phase_fits(int32_t B, int32_t items_per_wg, int64_t P);
int xrl_qmix_fused_update(const xrl_qmix_fused_t* p, xrl_stream_t stream);
int xrl_qmix_fused_lds_bytes(const xrl_qmix_fused_t* p);   /* LDS the launch needs (must be <= 160 KB), -1 on bad dims */
int xrl_qmix_fused_layout(const xrl_qmix_fused_t* p, xrl_qf_image_t* out);   /* needs n_layers, dims, N, S, H, HH only */

/* One acting step of the recurrent agents as ONE launch (value_factorization.py:66-92 with Basic_RNN, rnn.py:52-77: mlp
 * blocks -> nn.GRU cell -> Q head; the reference calls it once per vector step, off_policy_marl.py:478-486): Q values of R
 * rows from their observations and carried hidden states; the new hidden states replace the old ones.  rows_per_wg rows
 * per workgroup, weights from an IMAGE in the layout of xrl_marl_act_gru_layout (layers in the order
 * pre[0..n_pre), W_ih, W_hh, post[0..n_post); matrix l at w[l], row-major or interleaved: xrl_qa_image_t; its bias at b[l]; padding
 * zero) -- staged in LDS by the general kernel, held in registers (one thread per output) by the kernel for one row per workgroup.  Cell
 * arithmetic as xrl_gru_forward.  Same numbers as xrl_linear_fwd + xrl_gru_forward + xrl_linear_fwd up to fp32 summation
 * order.  H == 0: feed-forward agents (no recurrent layer: pre = every hidden layer, post = the output layer; h unused). */
#define XRL_QA_MAX_LAYERS 8
typedef struct {
    int32_t w[XRL_QA_MAX_LAYERS], b[XRL_QA_MAX_LAYERS], ldw[XRL_QA_MAX_LAYERS];
    int32_t image_floats, lds_bytes;
    int32_t interleaved;        /* 0: matrix l row-major, element (n, k) at w[l] + n * ldw[l] + k.  1 (the one-thread-per-output kernel:
                                 * rows_per_wg == 1, every layer <= 64 wide, H 0 or 64, O <= 64, lds_staged == 0): k-quads interleaved
                                 * over the outputs, element (n, k) at w[l] + ((k / 4) * ldw[l] + n) * 4 + k % 4, ldw = outputs padded to 64 */
    int32_t pad;
} xrl_qa_image_t;
typedef struct {
    const float* image;
    const float* obs;           /* [R][O] */
    float* h;                   /* [R][H] in: state before the step, out: state after it */
    const float* reset;         /* NULL or [R] f32: != 0 -> this row starts from the zero state (rnn.py:86-92) */
    float* q;                   /* [R][ldq] out */
    int32_t R, rows_per_wg, O, H, ldq, act;     /* act: XRL_ACT_* after every pre layer and every post layer but the last */
    int32_t n_pre, n_post;
    int32_t pre[3], post[3];    /* widths: pre = the mlp blocks below the GRU, post = Q head (last = n_actions) */
    /* optional: xrl_marl_select_actions for the same rows in the same launch (same arithmetic, same Philox keys) */
    int32_t* action;            /* NULL (no selection) or [R] */
    float* action_f;            /* NULL or [R] */
    const float* avail;         /* NULL or [R][n_actions] f32 0/1 */
    const float* eps_dev;       /* [1] with action, or NULL: `eps` below */
    const uint32_t* step_dev;   /* NULL or [1]: added to step */
    uint64_t seed;
    uint32_t step;
    float eps;                  /* used when eps_dev is NULL */
    int32_t lds_staged;         /* != 0: the kernel that stages a row-major image in LDS even where the other one could serve (bit-identical) */
    int32_t pad;
} xrl_marl_act_gru_t;
int xrl_marl_act_gru(const xrl_marl_act_gru_t* p, xrl_stream_t stream);
int xrl_marl_act_gru_layout(const xrl_marl_act_gru_t* p, xrl_qa_image_t* out);
/* Measurement aid (tools/probe_act_gru.py): stamps != NULL -> every workgroup of the following xrl_marl_act_gru launches writes 16
 * int64 clock stamps (slot 0 / 15: s_memrealtime at start / end, 1..7: s_memtime at the phase boundaries) to stamps[16 * workgroup];
 * NULL switches it off again.  Not for production use. */
int xrl_debug_act_gru_stamps(long long* stamps);

/* One-layer GRU over whole sequences, time-major (Basic_RNN, rl_models/representations/rnn.py:52-77; nn.GRU built by
 * rl_models/modules/layers.py:79-98; the recurrent agents of qmix/sc2/3m.yaml).  gi = x W_ih^T + b_ih for all steps is
 * the caller's GEMM (xrl_linear_fwd); this launch runs the serial part, one wavefront per sequence. H must be 64. */
typedef struct {
    const float* gi;      /* [T1][R][ld_gi]: input-side gate pre-activations r | z | n (3H values per row) */
    const float* w_hh;    /* [3H][H]  weight_hh_l0 */
    const float* b_hh;    /* [3H]     bias_hh_l0 */
    const float* h0;      /* NULL (zeros: init_rnn_states, rnn.py:79-84) or [R][H] */
    const float* reset;   /* NULL or [R] f32: != 0 -> start this row from zeros (init_rnn_states_item, rnn.py:86-92) */
    float* hs;            /* [T1+1][R][H]: slot 0 = initial state, slot t+1 = h_t */
    float* gates;         /* NULL or [T1][R][4H]: r | z | n | W_hn h + b_hn, kept for xrl_gru_backward */
    float* h_last;        /* NULL or [R][H]: h_{T1-1} (may alias h0: the state carried between acting steps) */
    int32_t R, T1, H, ld_gi;
    /* optional second problem of the same launch (same R, T1, zero initial state, no gates kept): the target network's
     * sequences of a QMIX update -- the recurrence is latency-bound, two problems cost the time of one.  gi2 NULL = none */
    const float* gi2; const float* w_hh2; const float* b_hh2; float* hs2;
} xrl_gru_fwd_t;
int xrl_gru_forward(const xrl_gru_fwd_t* p, xrl_stream_t stream);

/* Back-propagation through time of xrl_gru_forward.  d_gi feeds the backward pass of the layers below (and dW_ih, db_ih);
 * dW_hh = d_gh^T [hs slots 0..T1-1], db_hh = column sums of d_gh are the caller's GEMM (xrl_linear_bwd_weight). */
typedef struct {
    const float* d_hs;    /* [T1][R][ld_dhs] d loss / d h_t */
    const float* hs;      /* [T1+1][R][H] from the forward pass */
    const float* gates;   /* [T1][R][4H] from the forward pass */
    const float* w_hh;    /* [3H][H] */
    float* d_gi;          /* [T1][R][ld_dgi] */
    float* d_gh;          /* [T1][R][3H] */
    float* d_h0;          /* NULL or [R][H] */
    int32_t R, T1, H, ld_dhs, ld_dgi, pad;
} xrl_gru_bwd_t;
int xrl_gru_backward(const xrl_gru_bwd_t* p, xrl_stream_t stream);

/* One-layer LSTM, the `rnn: "LSTM"` option of Basic_RNN (rnn.py:45-47,59-66; layers.py:101-125): same conventions as the
 * GRU entry points above, gate order i | f | g | o (torch.nn.LSTM), 4H values per row of gi / gates / d_gates. */
typedef struct {
    const float* gi;      /* [T1][R][ld_gi] */
    const float* w_hh;    /* [4H][H] */
    const float* b_hh;    /* [4H] */
    const float* h0;      /* NULL (zeros) or [R][H] */
    const float* c0;      /* NULL (zeros) or [R][H] */
    const float* reset;   /* NULL or [R]: != 0 -> this row starts from zero hidden AND cell state */
    float* hs;            /* [T1+1][R][H] */
    float* cs;            /* NULL or [T1+1][R][H] cell states (slot 0 = initial), kept for xrl_lstm_backward */
    float* gates;         /* NULL or [T1][R][4H] activated gates i | f | g | o */
    float* h_last;        /* NULL or [R][H] (may alias h0) */
    float* c_last;        /* NULL or [R][H] (may alias c0) */
    int32_t R, T1, H, ld_gi;
    const float* gi2; const float* w_hh2; const float* b_hh2; float* hs2;   /* optional second problem (target network) */
} xrl_lstm_fwd_t;
int xrl_lstm_forward(const xrl_lstm_fwd_t* p, xrl_stream_t stream);
typedef struct {
    const float* d_hs;    /* [T1][R][ld_dhs] */
    const float* cs;      /* [T1+1][R][H] */
    const float* gates;   /* [T1][R][4H] */
    const float* w_hh;    /* [4H][H] */
    float* d_gates;       /* [T1][R][ld_dg]: gradient of the gate pre-activations (input AND hidden side) */
    int32_t R, T1, H, ld_dhs, ld_dg, pad;
} xrl_lstm_bwd_t;
int xrl_lstm_backward(const xrl_lstm_bwd_t* p, xrl_stream_t stream);

/* ------------------------------------------------------------------ episode replay buffer (recurrent multi-agent path)
 * MARL_OffPolicyBuffer_RNN (xuance/common/memory_tools_marl.py:770-996).  Every field is [episode][slots][row_bytes]
 * (slots = T for per-step fields, T+1 for obs / state / avail_actions); `a`, `b`, `c` per entry point: */
typedef struct {
    void* a;
    const void* b;
    const void* c;
    int32_t row_bytes;     /* bytes of one time slot (all agents of a step), multiple of 4 */
    int32_t slots;
    int32_t flags;         /* xrl_episode_finish: bit0 = zero the staging row after the copy (the `filled` field) */
    int32_t pad;
    const void* d;         /* xrl_episode_store_finish only: this step's data [n_envs][row] for the field, or NULL */
} xrl_episode_field_t;
/* Device-side loop control of run_episodes (off_policy_marl.py:464-546) for callers that enqueue vector steps ahead of the
 * host's knowledge of their outcome; launch once at the end of every (captured) vector step -- see csrc/episodes.hip. */
typedef struct {
    const int64_t* totals;     /* [2] the provider's running totals: episodes finished, env steps in them */
    const int64_t* base;       /* [2] the totals when the call began */
    const int64_t* call;       /* [2] the agent's current_step when the call began, n_episodes of the call */
    int64_t* snap;             /* [2] out: (episodes, env steps) of this call as of the last step that counted */
    int32_t* active;           /* [1] in: the step that just ran counted; out: the next one will (host sets 1 per call) */
    double* e_state;           /* [1] in/out: the agent's e_greedy (double, as the host keeps it) */
    float* eps_dev;            /* [1] out: e_greedy as the action-selection kernels read it */
    float* active_f;           /* [1] out: 1.0f / 0.0f -- multiply `done` with it before xrl_episode_finish */
    int32_t* active_i;         /* NULL or [2] out: 1 / 0 twice (for callers that advance their counters themselves) */
    int32_t* host_flags;       /* [ring] or NULL: device pointer of pinned host memory (xrl_host_device_pointer); launch k of a
                                * call writes 1 + its `active` output to host_flags[k % ring] with system scope (a host that
                                * zeroes a slot before the launch can poll it: 0 = not yet) */
    int32_t* seq;              /* [1] in/out: launches of this call so far (the host sets 0 per call); NULL iff host_flags NULL */
    double start_greedy, end_greedy, delta_greedy;
    int32_t ring;
    int32_t reset_rule;        /* which rows of reset_rows a finished env i flags: 0 = its own rows [i][0..n_agents); 1 = the reference's
                                * rule, flattened row i only (init_rnn_states_item passes batch_index = [i_env] to a state tensor whose
                                * batch axis is n_envs * n_agents: value_factorization.py:161-167, representations/rnn.py:86-92) */
    /* optional bookkeeping of the same step, done by the same launch */
    const float* done;         /* [n_envs] with reset_rows */
    float* reset_rows;         /* NULL or [n_envs][n_agents]: the rows that start their next step from zero state (reset_rule) */
    uint32_t* counters;        /* NULL or [2]: both advanced by `active` as it was when the step ran (RNG step counters) */
    int32_t n_envs, n_agents;
    int32_t* ptr_size;         /* NULL, or the episode ring's {ptr, size}: advanced here by the number of done envs when the
                                * step counted (then call xrl_episode_finish_gated with advance = 0) */
    int32_t buffer_size, pad2;
    const int32_t* end_step;   /* NULL, or [n_envs] info["episode_step"] of the envs that finished: the epsilon update is then applied
                                * once PER FINISHED ENV in env order, on the running current_step, as the reference's loop does
                                * (off_policy_marl.py:532-534) -- it differs from one application on the step's total only in the step
                                * where epsilon crosses end_greedy with more than one env finishing */
    /* NULL, or: stored_state [n_envs][state_dim] <- xrl_marl_stored_state(next_state, done) for the NEXT vector step (what the
     * reference's loop will store as its state: every row = the row of the last env that finished in THIS step, else a copy) --
     * the same launch instead of one of its own per captured step */
    const float* next_state;
    float* stored_state;
    int32_t state_dim, pad3;
} xrl_marl_gate_t;
/* Device address of page-locked host memory (hipHostMalloc / torch pin_memory), for kernels that publish a word to the host. */
int xrl_host_device_pointer(void* pinned_host, void** device_out);
int xrl_marl_loop_gate(const xrl_marl_gate_t* gate, xrl_stream_t stream);
/* xrl_episode_store_finish followed by xrl_marl_loop_gate as ONE launch (the captured vector step of run_episodes: store_experience +
 * finish_path of memory_tools_marl.py:904-968, then the loop control of off_policy_marl.py:464-546): every block of the finish takes a
 * ticket when it is done, the block that draws the last one runs the gate's statements -- after every read of `gate` (= loop_gate's
 * active_f) and of the ring's ptr that the gate rewrites.  ticket: [1] u32 of the caller, zero before the first launch (re-armed by
 * the launch).  Same results as the two launches. */
int xrl_episode_store_finish_gate(const xrl_episode_field_t* fields, int n_fields, const int32_t* steps, const float* gate,
                                  const float* done, const int32_t* end_step, const int32_t* ptr_size, int n_envs,
                                  int buffer_size, const xrl_marl_gate_t* loop_gate, uint32_t* ticket, xrl_stream_t stream);
/* out [n_envs][state_dim] = the global state OffPolicyMARLAgents.train / run_episodes STORE for the coming vector step
 * (core/off_policy_marl.py:384-399, 496-511 with store_experience :151-153 and memory_tools_marl.py:731-740): a copy of `state`
 * unless some env finished its episode in the previous step (done_prev [n_envs] != 0; NULL = none) -- then EVERY row is the
 * (reset) state row of the last such env, because the loops assign `state = info[i]["reset_state"]` (the list itself, not
 * element i) and the buffer broadcasts the single vector over its env axis.  A reference defect, reproduced so that buffers and
 * updates equal the reference's (QMIX_Agents: config.reference_state_broadcast, default on). */
int xrl_marl_stored_state(const float* state, const float* done_prev, float* out, int n_envs, int state_dim, xrl_stream_t stream);

/* store (:904-921): a = staging [n_envs][slots][row], b = step data [n_envs][row]: staging[env][steps[env]] <- b[env]
 * (`filled` is a field whose step data is ones). */
int xrl_episode_store_step(const xrl_episode_field_t* fields, int n_fields, const int32_t* steps, int n_envs,
                           xrl_stream_t stream);
/* finish_path + store_episodes (:923-968) for every env with done[env] != 0, in env order: c (NULL or [n_envs][row]) is
 * written at slot end_step[env] of the staging row b, the whole row is copied to ring row a[(ptr + rank) % buffer_size],
 * then ptr_size = {ptr, size} advance (device-resident: captured graphs and sampling kernels read them). */
int xrl_episode_finish(const xrl_episode_field_t* fields, int n_fields, const float* done, const int32_t* end_step,
                       int32_t* ptr_size, int n_envs, int buffer_size, xrl_stream_t stream);
/* xrl_episode_store_step (a = unused / ring, b = staging, d = step data) followed by xrl_episode_finish_gated without the
 * pointer advance, as ONE launch: per field, staging[env][steps[env]] <- d[env], then the episode close of the envs with
 * done != 0.  ptr_size is only read. */
int xrl_episode_store_finish(const xrl_episode_field_t* fields, int n_fields, const int32_t* steps, const float* gate,
                             const float* done, const int32_t* end_step, const int32_t* ptr_size, int n_envs, int buffer_size,
                             xrl_stream_t stream);
/* Same, switched by a device scalar: gate NULL or *gate != 0 -> as above; *gate == 0 -> nothing is closed (a dry step of a
 * loop that runs ahead of its stop condition, xrl_marl_loop_gate). */
int xrl_episode_finish_gated(const xrl_episode_field_t* fields, int n_fields, const float* gate, const float* done,
                             const int32_t* end_step, int32_t* ptr_size, int n_envs, int buffer_size, int advance,
                             xrl_stream_t stream);   /* advance 0: ptr_size is only read (the caller advances it later) */
/* sample (:970-996): a = time-major batch [slots][B][row], b = ring: a[t][i] <- b[idx[i]][t]. */
int xrl_episode_gather(const xrl_episode_field_t* fields, int n_fields, const int64_t* idx, int B, xrl_stream_t stream);
/* xrl_sample_replay_indices(idx_out, B, 1, n_size, ...) + xrl_episode_gather in ONE launch (same Philox stream, same episodes). */
int xrl_episode_gather_sampled(const xrl_episode_field_t* fields, int n_fields, int64_t* idx_out, int B, int n_size,
                               const int32_t* size_dev, uint64_t seed, uint32_t counter, const uint32_t* counter_dev,
                               xrl_stream_t stream);

/* ------------------------------------------------------------------ prioritized replay (PerOffPolicyBuffer)
 * memory_tools.py:471-598 with segtree_tool.py:24-230.  Per-env sum and min segment trees, [n_envs][2*capacity] float64
 * (capacity = power of two >= n_size; node 1 = root; unset sum leaves 0, unset min leaves +inf), max_priority [n_envs]. */
/* store (:536-538): leaf `ptr` of every env <- max_priority[env] ** alpha */
int xrl_per_store(double* sum_tree, double* min_tree, const double* max_priority, int ptr, double alpha, int n_envs,
                  int capacity, xrl_stream_t stream);
/* sample (:499-507, 542-565): per env `per_env` stratified draws, mass = (u + i) * sum(0, size-1) / per_env with the
 * caller's uniforms [n_envs][per_env] (random.random() in the reference), find_prefixsum_idx, importance weights.
 * flat_idx (optional) = env * n_size + step for xrl_soa_gather. */
int xrl_per_sample(const double* sum_tree, const double* min_tree, const double* uniforms, int size, double beta, int n_envs,
                   int n_size, int capacity, int per_env, int64_t* step_choices, double* weights, int64_t* flat_idx,
                   xrl_stream_t stream);
/* update_priorities (:586-597): per env, in order: p == 0 -> 1e-8; leaf <- p ** alpha; max_priority = max(., p) */
int xrl_per_update_priorities(double* sum_tree, double* min_tree, double* max_priority, const int64_t* idxes,
                              const float* priorities, double alpha, int n_envs, int capacity, int per_env,
                              xrl_stream_t stream);

/* Hard target update inside a captured graph: if (state->step % sync_frequency == 0) target <- params
 * (dqn_learner.py:56-57, qmix_learner.py:105-106; copy_target deep_q_network.py:95-99). */
int xrl_sync_target(const float* params, float* target, int64_t P, const xrl_adam_state_t* state,
                    int sync_frequency, xrl_stream_t stream);

/* ------------------------------------------------------------------ hipGraph capture of op sequences */
int xrl_graph_begin(xrl_stream_t stream);
int xrl_graph_end(xrl_stream_t stream, void** graph_exec_out);
int xrl_graph_launch(void* graph_exec, xrl_stream_t stream);
int xrl_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* XRL_HIP_H */
