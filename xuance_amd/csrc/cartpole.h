// CartPole-v1 dynamics (Barto, Sutton & Anderson 1983; constants and explicit-Euler update as published with Gymnasium's
// classic_control/cartpole.py -- not part of the reference tree).  float64 state like Gymnasium.
#pragma once
#include "rng.h"

namespace xrl {

__device__ __forceinline__ void cartpole_reset(double* s, uint64_t seed, int e, uint32_t episode) {
    uint32_t r[4], q[4];
    philox4x32(seed, (uint32_t)e, episode, STREAM_RESET_A, r);
    philox4x32(seed, (uint32_t)e, episode, STREAM_RESET_B, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = -0.05 + 0.1 * u01d(r[j], q[j]);   // uniform(-0.05, 0.05)
}

// sin / cos of a pole angle.  The angle of a live episode stays below 0.21 rad (+ one step), so the argument never needs a range
// reduction: |th| <= pi/4 runs the two polynomial kernels every libm ends in (coefficients and evaluation order of fdlibm's
// __kernel_sin / __kernel_cos, Sun Microsystems, public domain; < 1 ulp), anything else the library's sincos.  The library
// routine spends ~3 k cycles per call on its argument reduction paths -- it used to hide under the 4 k-cycle MFMA chain of the
// 32-row rollout tiles and was the longest wave of a vector step once that chain had been halved.
__device__ __forceinline__ void pole_sincos(double x, double* s, double* c) {
#pragma clang fp contract(off)
    if (fabs(x) <= 0.78125) {
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        const double z = x * x, v = z * x;
        const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        *s = x + v * (S1 + z * rs);
        const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        if (fabs(x) < 0.3) {
            *c = 1.0 - (0.5 * z - z * rc);
        } else {
            const double qx = __hiloint2double(__double2hiint(fabs(x)) - 0x00200000, 0);      // ~ |x| / 4
            const double hz = 0.5 * z - qx, a = 1.0 - qx;
            *c = a - (hz - z * rc);
        }
    } else {
        sincos(x, s, c);
    }
}

// one physics step from state s with action a (0/1): new state in (x, xd, th, thd), terminated flag
__device__ __forceinline__ void cartpole_advance(const double* s, int a, double& x, double& xd, double& th, double& thd,
                                                 bool& term) {
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
    const double total_mass = masspole + masscart, polemass_length = masspole * length;
    const double theta_thr = 12.0 * 2.0 * 3.14159265358979323846 / 360.0, x_thr = 2.4;
    x = s[0]; xd = s[1]; th = s[2]; thd = s[3];
    const double force = a == 1 ? force_mag : -force_mag;
    double st, ct;
    pole_sincos(th, &st, &ct);
    const double temp = (force + polemass_length * thd * thd * st) / total_mass;
    const double thacc = (gravity * st - ct * temp) / (length * (4.0 / 3.0 - masspole * ct * ct / total_mass));
    const double xacc = temp - polemass_length * thacc * ct / total_mass;
    x = x + tau * xd; xd = xd + tau * xacc; th = th + tau * thd; thd = thd + tau * thacc;   // explicit Euler
    term = (x < -x_thr) || (x > x_thr) || (th < -theta_thr) || (th > theta_thr);
}

// One env of xrl_cartpole_step (cartpole_step_kernel's statements; also csrc/act_tail.hip)
__device__ __forceinline__ void cartpole_step_one(const xrl_cartpole_t& p, int e) {
    double* s = p.state + (size_t)e * 4;
    double x, xd, th, thd;
    bool term;
    cartpole_advance(s, p.action[e], x, xd, th, thd, term);
    const int steps = p.steps[e] + 1;
    const bool trunc = steps >= p.max_steps;
    float* no = p.next_obs + (size_t)e * 4;
    no[0] = (float)x; no[1] = (float)xd; no[2] = (float)th; no[3] = (float)thd;
    p.reward[e] = 1.0f;
    p.terminated[e] = term ? 1.f : 0.f;
    p.truncated[e] = trunc ? 1.f : 0.f;
    const float score = p.ep_score[e] + 1.0f;
    float* o = p.obs + (size_t)e * 4;
    if (term || trunc) {
        const int ep = p.episodes[e] + 1;
        p.episodes[e] = ep;
        cartpole_reset(s, p.seed, e, (uint32_t)ep);
        p.steps[e] = 0;
        p.ep_score[e] = 0.f;
        o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = (float)s[2]; o[3] = (float)s[3];   // info["reset_obs"]
        atomicAdd(&p.stats[0], 1.0); atomicAdd(&p.stats[1], (double)score); atomicAdd(&p.stats[2], (double)steps);
    } else {
        s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
        p.steps[e] = steps;
        p.ep_score[e] = score;
        o[0] = no[0]; o[1] = no[1]; o[2] = no[2]; o[3] = no[3];
    }
}

}  // namespace xrl
