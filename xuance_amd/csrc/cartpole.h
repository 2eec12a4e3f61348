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

// one physics step from state s with action a (0/1): new state in (x, xd, th, thd), terminated flag
__device__ __forceinline__ void cartpole_advance(const double* s, int a, double& x, double& xd, double& th, double& thd,
                                                 bool& term) {
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
    const double total_mass = masspole + masscart, polemass_length = masspole * length;
    const double theta_thr = 12.0 * 2.0 * 3.14159265358979323846 / 360.0, x_thr = 2.4;
    x = s[0]; xd = s[1]; th = s[2]; thd = s[3];
    const double force = a == 1 ? force_mag : -force_mag;
    double st, ct;
    sincos(th, &st, &ct);
    const double temp = (force + polemass_length * thd * thd * st) / total_mass;
    const double thacc = (gravity * st - ct * temp) / (length * (4.0 / 3.0 - masspole * ct * ct / total_mass));
    const double xacc = temp - polemass_length * thacc * ct / total_mass;
    x = x + tau * xd; xd = xd + tau * xacc; th = th + tau * thd; thd = thd + tau * thacc;   // explicit Euler
    term = (x < -x_thr) || (x > x_thr) || (th < -theta_thr) || (th > theta_thr);
}

}  // namespace xrl
