// Pendulum-v1, MountainCar-v0 and Acrobot-v1 dynamics as published with Gymnasium's classic_control package (pendulum.py,
// mountain_car.py, acrobot.py; pinned by the reference as gymnasium >= 0.28, < 1.3 in setup.py:73 -- third-party, NOT in the
// reference tree and not in this image: the equations and constants below restate the published ones, oracle/xrl_oracle.py holds
// the NumPy twin).  float64 state like Gymnasium, float32 observations; initial states from the engine's Philox streams.
#pragma once
#include "rng.h"

namespace xrl {

enum { CLASSIC_PENDULUM = 1, CLASSIC_MOUNTAINCAR = 2, CLASSIC_ACROBOT = 3 };

constexpr double CL_PI = 3.14159265358979323846;

// uniform initial state of env e at the start of its `episode`-th episode
__device__ __forceinline__ void classic_reset(int kind, double* s, uint64_t seed, int e, uint32_t episode) {
    uint32_t r[4], q[4];
    philox4x32(seed, (uint32_t)e, episode, STREAM_RESET_A, r);
    philox4x32(seed, (uint32_t)e, episode, STREAM_RESET_B, q);
    double u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = u01d(r[j], q[j]);
    if (kind == CLASSIC_PENDULUM) {                 // uniform(-[pi, 1], [pi, 1])                      (pendulum.py: reset)
        s[0] = -CL_PI + 2.0 * CL_PI * u[0]; s[1] = -1.0 + 2.0 * u[1]; s[2] = 0.0; s[3] = 0.0;
    } else if (kind == CLASSIC_MOUNTAINCAR) {       // position uniform(-0.6, -0.4), velocity 0          (mountain_car.py: reset)
        s[0] = -0.6 + 0.2 * u[0]; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    } else {                                        // uniform(-0.1, 0.1) for all four                   (acrobot.py: reset)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = -0.1 + 0.2 * u[j];
    }
}

__device__ __forceinline__ int classic_obs_dim(int kind) { return kind == CLASSIC_PENDULUM ? 3 : (kind == CLASSIC_MOUNTAINCAR ? 2 : 6); }

__device__ __forceinline__ void classic_observe(int kind, const double* s, float* o) {
    if (kind == CLASSIC_PENDULUM) { o[0] = (float)cos(s[0]); o[1] = (float)sin(s[0]); o[2] = (float)s[1]; }
    else if (kind == CLASSIC_MOUNTAINCAR) { o[0] = (float)s[0]; o[1] = (float)s[1]; }
    else { o[0] = (float)cos(s[0]); o[1] = (float)sin(s[0]); o[2] = (float)cos(s[1]); o[3] = (float)sin(s[1]); o[4] = (float)s[2]; o[5] = (float)s[3]; }
}

// acrobot.py: _dsdt ("book" variant, torque a), state (theta1, theta2, dtheta1, dtheta2)
__device__ __forceinline__ void acrobot_dsdt(const double* s, double a, double* d) {
    const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8;
    const double t1 = s[0], t2 = s[1], dt1 = s[2], dt2 = s[3];
    const double d1 = m1 * lc1 * lc1 + m2 * (l1 * l1 + lc2 * lc2 + 2.0 * l1 * lc2 * cos(t2)) + I1 + I2;
    const double d2 = m2 * (lc2 * lc2 + l1 * lc2 * cos(t2)) + I2;
    const double phi2 = m2 * lc2 * g * cos(t1 + t2 - CL_PI / 2.0);
    const double phi1 = -m2 * l1 * lc2 * dt2 * dt2 * sin(t2) - 2.0 * m2 * l1 * lc2 * dt2 * dt1 * sin(t2) +
                        (m1 * lc1 + m2 * l1) * g * cos(t1 - CL_PI / 2.0) + phi2;
    const double ddt2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * dt1 * dt1 * sin(t2) - phi2) / (m2 * lc2 * lc2 + I2 - d2 * d2 / d1);
    const double ddt1 = -(d2 * ddt2 + phi1) / d1;
    d[0] = dt1; d[1] = dt2; d[2] = ddt1; d[3] = ddt2;
}

__device__ __forceinline__ double classic_wrap(double x, double m, double M) {   // acrobot.py: wrap
    const double diff = M - m;
    while (x > M) x -= diff;
    while (x < m) x += diff;
    return x;
}

// One step from state s: new state in ns, reward, terminated.  ai: discrete action, af: continuous action (Pendulum).
__device__ __forceinline__ void classic_advance(int kind, const double* s, int ai, float af, double* ns, float& reward, bool& term) {
    ns[0] = s[0]; ns[1] = s[1]; ns[2] = s[2]; ns[3] = s[3];
    if (kind == CLASSIC_PENDULUM) {
        const double max_speed = 8.0, max_torque = 2.0, dt = 0.05, g = 10.0, m = 1.0, l = 1.0;
        const float uc = fminf(fmaxf(af, (float)-max_torque), (float)max_torque);        // np.clip on the float32 action
        const double u = (double)uc, th = s[0], thdot = s[1];
        double an = fmod(th + CL_PI, 2.0 * CL_PI);                                        // angle_normalize: ((x + pi) % (2 pi)) - pi
        if (an < 0.0) an += 2.0 * CL_PI;                                                  //   (Python's % takes the divisor's sign)
        an -= CL_PI;
        const double costs = an * an + 0.1 * thdot * thdot + 0.001 * (u * u);
        double nthdot = thdot + (3.0 * g / (2.0 * l) * sin(th) + 3.0 / (m * l * l) * u) * dt;
        nthdot = fmin(fmax(nthdot, -max_speed), max_speed);
        ns[0] = th + nthdot * dt; ns[1] = nthdot;
        reward = (float)(-costs); term = false;
    } else if (kind == CLASSIC_MOUNTAINCAR) {
        const double min_position = -1.2, max_position = 0.6, max_speed = 0.07, goal_position = 0.5, force = 0.001, gravity = 0.0025;
        double position = s[0], velocity = s[1];
        velocity += (double)(ai - 1) * force + cos(3.0 * position) * (-gravity);
        velocity = fmin(fmax(velocity, -max_speed), max_speed);
        position += velocity;
        position = fmin(fmax(position, min_position), max_position);
        if (position == min_position && velocity < 0.0) velocity = 0.0;
        ns[0] = position; ns[1] = velocity;
        term = position >= goal_position && velocity >= 0.0;
        reward = -1.0f;
    } else {
        const double dt = 0.2, torque = (double)(ai - 1);                                 // AVAIL_TORQUE = [-1, 0, +1], no noise
        double k1[4], k2[4], k3[4], k4[4], y[4];                                          // acrobot.py: rk4 over [0, dt], one step
        acrobot_dsdt(s, torque, k1);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = s[j] + dt / 2.0 * k1[j];
        acrobot_dsdt(y, torque, k2);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = s[j] + dt / 2.0 * k2[j];
        acrobot_dsdt(y, torque, k3);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = s[j] + dt * k3[j];
        acrobot_dsdt(y, torque, k4);
#pragma unroll
        for (int j = 0; j < 4; ++j) ns[j] = s[j] + dt / 6.0 * (k1[j] + 2.0 * k2[j] + 2.0 * k3[j] + k4[j]);
        ns[0] = classic_wrap(ns[0], -CL_PI, CL_PI);
        ns[1] = classic_wrap(ns[1], -CL_PI, CL_PI);
        ns[2] = fmin(fmax(ns[2], -4.0 * CL_PI), 4.0 * CL_PI);
        ns[3] = fmin(fmax(ns[3], -9.0 * CL_PI), 9.0 * CL_PI);
        term = -cos(ns[0]) - cos(ns[1] + ns[0]) > 1.0;
        reward = term ? 0.f : -1.f;
    }
}

// One env of xrl_classic_step (classic_step_kernel's statements; also the tail launch of the on-policy acting step, csrc/act_tail.hip)
__device__ __forceinline__ void classic_step_one(const xrl_classic_t& p, int e) {
    const int kind = p.kind, D = classic_obs_dim(kind);
    double* s = p.state + (size_t)e * 4;
    float* o = p.obs + (size_t)e * D;
    double ns[4];
    float reward;
    bool term;
    classic_advance(kind, s, p.action ? p.action[e] : 0, p.action_f ? p.action_f[e] : 0.f, ns, reward, term);
    const int steps = p.steps[e] + 1;
    const bool trunc = steps >= p.max_steps;
    float* no = p.next_obs + (size_t)e * D;
    classic_observe(kind, ns, no);
    p.reward[e] = reward;
    p.terminated[e] = term ? 1.f : 0.f;
    p.truncated[e] = trunc ? 1.f : 0.f;
    const float score = p.ep_score[e] + reward;
    if (term || trunc) {
        const int ep = p.episodes[e] + 1;
        p.episodes[e] = ep;
        classic_reset(kind, s, p.seed, e, (uint32_t)ep);
        p.steps[e] = 0;
        p.ep_score[e] = 0.f;
        classic_observe(kind, s, o);                                     // info["reset_obs"]
        atomicAdd(&p.stats[0], 1.0); atomicAdd(&p.stats[1], (double)score); atomicAdd(&p.stats[2], (double)steps);
    } else {
        s[0] = ns[0]; s[1] = ns[1]; s[2] = ns[2]; s[3] = ns[3];
        p.steps[e] = steps;
        p.ep_score[e] = score;
        for (int j = 0; j < D; ++j) o[j] = no[j];
    }
}

}  // namespace xrl
