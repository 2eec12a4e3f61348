// PPO-clip loss head: log-prob / entropy / ratio / clipped surrogate / value MSE and their gradients with
// respect to the actor-head output and the value, one thread per sample (HBM-bound, ~50 B per sample).
// Reference arithmetic: xuance/torch/learners/policy_gradient/ppo_learner.py:46-60,70 with
// CategoricalDistribution / DiagGaussianDistribution (rl_models/modules/distributions.py:128-192) and the
// autograd rules of torch.clamp (gradient where lo <= x <= hi) and torch.minimum (0.5/0.5 on ties).
#include "common.h"
#include "ppo_math.h"

namespace xrl {

constexpr int LOSS_THREADS = 256;

template <bool GAUSSIAN>
__global__ void __launch_bounds__(LOSS_THREADS) ppo_loss_kernel(xrl_ppo_loss_t p) {
    __shared__ double scratch[16];
    const int chunk = (p.M + p.n_split - 1) / p.n_split;
    const int mbeg = blockIdx.x * chunk, mend = min(p.M, mbeg + chunk);
    const float invM = 1.f / (float)p.M;
    const float lo = (float)(1.0 - (double)p.clip_range), hi = (float)(1.0 + (double)p.clip_range);
    float mean = 0.f, denom = 1.f;
    if (p.stats) { mean = p.stats[0]; denom = p.stats[1] + 1e-8f; }
    const int A = p.A;

    double acc_s = 0.0, acc_c = 0.0, acc_e = 0.0, acc_v = 0.0, acc_n = 0.0, acc_k = 0.0;
    const bool klm = p.mode == 3;                        // PPOKL_Learner (ppokl_learner.py:35-101)
    const float klc = klm ? (float)*p.kl_coef : 0.f;     // (a Python float times a float32 tensor in the reference)
    // gaussian: per-thread partial of d log_std (A <= 32)
    float dls[32];
    if (GAUSSIAN) {
#pragma unroll
        for (int j = 0; j < 32; ++j) dls[j] = 0.f;
    }

    for (int m = mbeg + threadIdx.x; m < mend; m += blockDim.x) {
        const float* o = p.out + (size_t)m * p.ld_out;
        const bool pg = p.mode == 2;                     // PG_Learner: weight = returns, no critic (pg_learner.py:40-45)
        float adv = pg ? p.returns[m] : p.adv[m];
        if (p.stats && !pg) adv = __fdiv_rn(__fsub_rn(adv, mean), denom);
        const float ret = p.returns[m], v = pg ? 0.f : p.value[(size_t)m * p.ld_v], oldlp = p.mode == 0 ? p.old_logp[m] : 0.f;
        float logp, ent;
        if (!GAUSSIAN) {
            const int a = (int)p.actions[m];
            float mx = o[0];
            for (int j = 1; j < A; ++j) mx = fmaxf(mx, o[j]);
            float se = 0.f;
            for (int j = 0; j < A; ++j) se += expf(o[j] - mx);
            const float lse = mx + logf(se);
            logp = o[a] - lse;
            ent = 0.f;
            for (int j = 0; j < A; ++j) { const float l = o[j] - lse; ent -= expf(l) * l; }
            float kl_row = 0.f, lse_o = 0.f, old_lp = oldlp;
            const float* qo = klm ? p.old_a + (size_t)m * A : nullptr;
            if (klm) {                                   // old distribution = Categorical(logits = stored logits): normalise,
                float mo = qo[0];                        // old log-prob of the action, KL(new || old) = sum p (l - q)
                for (int j = 1; j < A; ++j) mo = fmaxf(mo, qo[j]);
                float so = 0.f;
                for (int j = 0; j < A; ++j) so += expf(qo[j] - mo);
                lse_o = mo + logf(so);
                old_lp = qo[a] - lse_o;
                for (int j = 0; j < A; ++j) { const float l = o[j] - lse; kl_row += expf(l) * (l - (qo[j] - lse_o)); }
                acc_k += kl_row;
            }
            const Surrogate s = klm ? surrogate_kl(logp, old_lp, adv, invM)
                                    : (p.mode != 0 ? surrogate_a2c(logp, adv, invM) : surrogate(logp, oldlp, adv, lo, hi, invM));
            float* dq = p.d_out + (size_t)m * p.ld_out;
            const float ce = p.ent_coef * invM, ck = klc * invM;
            for (int j = 0; j < A; ++j) {
                const float l = o[j] - lse, pj = expf(l);
                // d logp/d z_j = 1[j==a] - p_j ;  d H/d z_j = -p_j (l_j + H) ; loss has  -ent_coef * mean(H)
                float g = s.dlogp * ((j == a ? 1.f : 0.f) - pj) + ce * pj * (l + ent);
                if (klm) g += ck * pj * (l - (qo[j] - lse_o) - kl_row);          // d kl_row / d z_j
                dq[j] = g;
            }
            acc_s += (double)fminf(s.s1, s.s2); acc_n += s.clipped;
            if (p.diag) { p.diag[m] = logp; p.diag[p.M + m] = s.ratio; p.diag[2 * (size_t)p.M + m] = s.s1; p.diag[3 * (size_t)p.M + m] = s.s2; }
        } else {
            const float* x = p.actions + (size_t)m * A;
            logp = 0.f; ent = 0.f;
            for (int j = 0; j < A; ++j) {
                const float ls = p.log_std[j], sd = expf(ls), var = sd * sd, df = x[j] - o[j];
                logp += -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;     // Normal.log_prob, summed (:179-180)
                ent += 0.5f + LOG_SQRT_2PI + logf(sd);                           // Normal.entropy, summed (:182-183)
            }
            float old_lp = oldlp;
            const float* mo = klm ? p.old_a + (size_t)m * A : nullptr;
            const float* so = klm ? p.old_b + (size_t)m * A : nullptr;
            if (klm) {                                   // old distribution = Normal(old mu, old std): its log-prob of the action
                old_lp = 0.f;
                for (int j = 0; j < A; ++j) {
                    const float d0 = x[j] - mo[j];
                    old_lp += -(d0 * d0) / (2.f * so[j] * so[j]) - logf(so[j]) - LOG_SQRT_2PI;
                }
            }
            const Surrogate s = klm ? surrogate_kl(logp, old_lp, adv, invM)
                                    : (p.mode != 0 ? surrogate_a2c(logp, adv, invM) : surrogate(logp, oldlp, adv, lo, hi, invM));
            float* dq = p.d_out + (size_t)m * p.ld_out;
            const float ck = klm ? klc * invM / (float)A : 0.f;                  // kl.mean() runs over rows x dims (torch's
#pragma unroll                                                                   // Normal-Normal KL is elementwise)
            for (int j = 0; j < 32; ++j) {
                if (j < A) {
                    const float ls = p.log_std[j], sd = expf(ls), var = sd * sd, df = x[j] - o[j];
                    float gmu = s.dlogp * df / var, gls = s.dlogp * (df * df / var - 1.f);
                    if (klm) {
                        const float vr = (sd / so[j]) * (sd / so[j]), dm = (o[j] - mo[j]) / so[j];
                        acc_k += 0.5f * (vr + dm * dm - 1.f - logf(vr));
                        gmu += ck * (o[j] - mo[j]) / (so[j] * so[j]);
                        gls += ck * (vr - 1.f);
                    }
                    dq[j] = gmu * act_grad_from_out(o[j], p.out_act);           // through activation_action
                    dls[j] += gls;
                }
            }
            acc_s += (double)fminf(s.s1, s.s2); acc_n += s.clipped;
            if (p.diag) { p.diag[m] = logp; p.diag[p.M + m] = s.ratio; p.diag[2 * (size_t)p.M + m] = s.s1; p.diag[3 * (size_t)p.M + m] = s.s2; }
        }
        acc_e += ent;
        if (!pg) {
            const float dv = v - ret;
            p.d_value[(size_t)m * p.ld_v] = p.vf_coef * 2.f * dv * invM;       // d(vf * mean((v-ret)^2))/dv
            acc_c += (double)dv * dv; acc_v += v;
        }
    }

    const double t0 = block_sum(acc_s, scratch), t1 = block_sum(acc_c, scratch), t2 = block_sum(acc_e, scratch),
                 t3 = block_sum(acc_v, scratch), t4 = block_sum(acc_n, scratch), t5 = block_sum(acc_k, scratch);
    if (threadIdx.x == 0) {
        double* q = p.partials + (size_t)blockIdx.x * 8;
        q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; q[4] = t4; q[5] = t5; q[6] = 0; q[7] = 0;
    }
    if (GAUSSIAN && p.d_log_std) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j < A) {
                double t = block_sum((double)dls[j], scratch);
                // d(-ent_coef * mean_m sum_j(log_std_j + c))/d log_std_j = -ent_coef, added once (slab 0)
                if (blockIdx.x == 0) t -= (double)p.ent_coef;
                if (threadIdx.x == 0) p.d_log_std[(size_t)blockIdx.x * p.slab_stride + j] = (float)t;
            }
        }
    }
}

// kl_coef schedule of PPOKL_Learner.update (ppokl_learner.py:62-66) on the device, so that chained updates need no host
// round trip: kl = (sum of the loss launch's kl partials) / count (float32, as the reference's tensor), compared with
// float32(target * 1.5) and float32(target * 0.5); the coefficient is a double like the reference's Python float.
__global__ void ppokl_adapt_kernel(const double* __restrict__ partials, int n_split, double count, double* kl_coef, double target_kl,
                                   float* kl_out) {
    double s = 0.0;
    for (int i = 0; i < n_split; ++i) s += partials[(size_t)i * 8 + 5];
    const float kl = (float)(s / count);
    double c = *kl_coef;
    kl_coef[1] = c;                                   // the coefficient the loss of THIS update was formed with (host log: actor-loss)
    if (kl > (float)(target_kl * 1.5)) c = c * 2.0;
    else if (kl < (float)(target_kl * 0.5)) c = c / 2.0;
    c = c < 0.1 ? 0.1 : (c > 20.0 ? 20.0 : c);
    *kl_coef = c;
    if (kl_out) *kl_out = kl;
}

__global__ void __launch_bounds__(64) sum_partials_kernel(const double* __restrict__ partials, int n_rows, int width,
                                                          double* __restrict__ out, long in_stride, long out_stride) {
    const int j = threadIdx.x;
    if (j >= width) return;
    partials += (size_t)blockIdx.x * in_stride;       // one independent sum per block (xrl_sum_partials_batched)
    out += (size_t)blockIdx.x * out_stride;
    double s = 0.0;
    int r = 0;
    for (; r + 8 <= n_rows; r += 8) {                 // loads issued 8 at a time; the summation order stays row by row
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = partials[(size_t)(r + q) * width + j];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; r < n_rows; ++r) s += partials[(size_t)r * width + j];
    out[j] = s;
}

// Many rows (the recurrent QMIX update has T*B = 1920): 128 row chunks x 8 columns, fixed-order tree over the chunks.
__global__ void __launch_bounds__(1024) sum_partials_wide_kernel(const double* __restrict__ partials, int n_rows, int width,
                                                                 double* __restrict__ out, long in_stride, long out_stride) {
    __shared__ double sh[1024];
    partials += (size_t)blockIdx.x * in_stride;
    out += (size_t)blockIdx.x * out_stride;
    const int j = threadIdx.x & 7, c = threadIdx.x >> 3;                     // width <= 8 here
    const int per = (n_rows + 127) / 128;
    double s = 0.0;
    if (j < width)
        for (int r = c * per; r < min(n_rows, (c + 1) * per); ++r) s += partials[(size_t)r * width + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int half = 64; half >= 1; half >>= 1) {
        if (c < half) sh[threadIdx.x] += sh[threadIdx.x + half * 8];
        __syncthreads();
    }
    if (c == 0 && j < width) out[j] = sh[j];
}

static int check(const xrl_ppo_loss_t* p, bool gaussian) {
    XRL_CHECK_ARG(p != nullptr);
    XRL_CHECK_ARG(p->mode >= 0 && p->mode <= 3);
    XRL_CHECK_ARG(p->out && p->actions && p->returns && p->d_out && p->partials && (p->old_logp || p->mode != 0));
    XRL_CHECK_ARG(p->mode == 2 || (p->value && p->adv && p->d_value));
    XRL_CHECK_ARG(p->M > 0 && p->A > 0 && p->A <= (gaussian ? 32 : 4096) && p->ld_out >= p->A && p->ld_v >= 1);
    XRL_CHECK_ARG(p->n_split >= 1);
    if (gaussian) XRL_CHECK_ARG(p->log_std != nullptr);
    if (p->mode == 3) XRL_CHECK_ARG(p->old_a && p->kl_coef && (!gaussian || p->old_b));
    return XRL_OK;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_ppo_loss_categorical(const xrl_ppo_loss_t* p, xrl_stream_t stream) {
    int rc = check(p, false);
    if (rc) return rc;
    hipLaunchKernelGGL(ppo_loss_kernel<false>, dim3(p->n_split), dim3(LOSS_THREADS), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_ppo_loss_gaussian(const xrl_ppo_loss_t* p, xrl_stream_t stream) {
    int rc = check(p, true);
    if (rc) return rc;
    hipLaunchKernelGGL(ppo_loss_kernel<true>, dim3(p->n_split), dim3(LOSS_THREADS), 0, as_stream(stream), *p);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_ppokl_adapt(const double* partials, int n_split, double count, double* kl_coef, double target_kl, float* kl_out,
                               xrl_stream_t stream) {
    XRL_CHECK_ARG(partials && kl_coef && n_split >= 1 && count > 0);
    hipLaunchKernelGGL(ppokl_adapt_kernel, dim3(1), dim3(1), 0, as_stream(stream), partials, n_split, count, kl_coef, target_kl, kl_out);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_sum_partials_batched(const double* partials, int n_rows, int width, double* out, int n_batches,
                                        long in_stride, long out_stride, xrl_stream_t stream) {
    XRL_CHECK_ARG(partials && out && n_rows > 0 && width > 0 && width <= 64 && n_batches > 0);
    XRL_CHECK_ARG(n_batches == 1 || (in_stride >= (long)n_rows * width && out_stride >= width));
    if (n_rows > 256 && width <= 8)
        hipLaunchKernelGGL(sum_partials_wide_kernel, dim3(n_batches), dim3(1024), 0, as_stream(stream), partials, n_rows, width,
                           out, in_stride, out_stride);
    else
        hipLaunchKernelGGL(sum_partials_kernel, dim3(n_batches), dim3(64), 0, as_stream(stream), partials, n_rows, width, out,
                           in_stride, out_stride);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_sum_partials(const double* partials, int n_rows, int width, double* out, xrl_stream_t stream) {
    return xrl_sum_partials_batched(partials, n_rows, width, out, 1, 0, 0, stream);
}
