"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference hot path.

This file is the *checker* for the HIP path in ``xuance_amd/``; it is never the
thing shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.

It restates, in plain NumPy with hand-derived backward passes (no autograd, no
import of the reference), the arithmetic of agi-brain/xuance v1.4.4's torch
backend for:

  * rollout buffer            xuance/common/memory_tools.py:182-287 (on-policy),
                              :331-387 (off-policy), :601-630 (Atari uint8)
  * running mean/std          xuance/common/statistic_tools.py:65-185,
                              xuance/torch/agents/base/agent.py:262-294
  * PPO-clip learner          xuance/torch/learners/policy_gradient/ppo_learner.py:35-95
  * DQN learner               xuance/torch/learners/qlearning_family/dqn_learner.py:28-75
  * QMIX learner (FF)         xuance/torch/learners/multi_agent_rl/qmix_learner.py:24-112,
                              iql_learner.py:37-83, q_mix_head.py:66-95
  * networks                  rl_models/modules/layers.py:16-33 (mlp_block),
                              heads/actor_head.py:14-72, heads/critic_head.py:9-30,
                              heads/q_head.py:11-39, modules/distributions.py:128-192
  * optimiser                 torch.optim.Adam(eps=1e-5) + clip_grad_norm_ + LinearLR
                              (PyTorch >=2.0,<3.0 -- third-party, restated from its
                              documented algorithm; call sites ppo_learner.py:18-22,61-67)

Pinning: every function here is checked against fixtures produced by running
the *unmodified reference* in the build container (oracle/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_vs_golden.py).  The reference's own
tests hold no numeric assertion for this path, so those
generated fixtures are the only pin.

All arithmetic is float32 unless ``dtype=np.float64`` is requested (used by
tests to bound rounding error).
"""
import math
import numpy as np

EPS = 1e-8  # xuance/common/common_tools.py EPS (used by agent.py:262-294)


# --------------------------------------------------------------------------------------
# Rollout buffer (on-policy)                                   memory_tools.py:182-287
# --------------------------------------------------------------------------------------
def gae_finish_path(rewards, values, dones, val, gamma, lam, use_gae=True):
    """One ``finish_path`` slice (memory_tools.py:242-265).

    Dtype quirk that parity has to reproduce (NumPy >= 2 promotion rules, the NumPy in this image):
    ``vs = np.append(values, [val])`` is float64 when ``val`` is a Python float -- which is what the
    agent passes for a *terminated* episode (``finish_path(0.0, i)``, ppo_agent.py:133,154) -- and
    float32 when ``val`` is a NumPy float32 (``vals[i]``: truncation / buffer-full bootstrap).  In the
    float64 case the recurrence is carried in float64 (with float32-rounded coefficients
    ``(1-d)*gamma`` and ``(1-d)*gamma*lam``) and only the stored advantages/returns are rounded to
    float32; in the float32 case everything is float32.  The expressions below are written with the
    same operand order as the reference so that NumPy applies the same promotions.
    (Under NumPy < 2, which the reference pins, scalar arithmetic with Python floats is float64
    everywhere; the difference to either branch here is <= 1 ulp of float32 per step.)

    rewards/values/dones: 1-D float32 arrays of the path.  Returns (returns, advantages) float32.
    """
    gamma, lam = float(gamma), float(lam)      # Python floats in the reference (config values): "weak" scalars
    rewards = np.asarray(rewards, np.float32)
    dones = np.asarray(dones, np.float32)
    vs = np.append(np.asarray(values, np.float32), [val], axis=0)        # :247
    L = len(rewards)
    if use_gae:
        adv = np.zeros_like(rewards)
        last = 0
        for t in reversed(range(L)):
            delta = rewards[t] + (1 - dones[t]) * gamma * vs[t + 1] - vs[t]          # :255
            adv[t] = last = delta + (1 - dones[t]) * gamma * lam * last             # :256
        ret = adv + vs[:-1]                                                        # :257
    else:
        # discount_cumsum == scipy.signal.lfilter([1],[1,-gamma], x[::-1])[::-1]  (common_tools.py:160-174)
        r = np.append(rewards, [val], axis=0)                                      # :259
        out = np.zeros(len(r), np.float64)
        acc = 0.0
        for t in reversed(range(len(r))):
            acc = float(r[t]) + gamma * acc
            out[t] = acc
        ret = out[:-1]                                                             # :260
        adv = r[:-1] + gamma * vs[1:] - vs[:-1]                                    # :261
    return np.asarray(ret).astype(np.float32), np.asarray(adv).astype(np.float32)


class OnPolicyBufferOracle:
    """Env-major restatement of DummyOnPolicyBuffer (memory_tools.py:182-287)."""

    def __init__(self, obs_shape, act_shape, n_envs, horizon_size, use_gae=True, use_advnorm=True,
                 gamma=0.99, gae_lam=0.95, obs_dtype=np.float32):
        self.obs_shape, self.act_shape = tuple(obs_shape), tuple(act_shape)
        self.n_envs, self.n_size = n_envs, horizon_size
        self.buffer_size = n_envs * horizon_size
        self.use_gae, self.use_advnorm = use_gae, use_advnorm
        self.gamma, self.gae_lam = gamma, gae_lam
        self.obs_dtype = obs_dtype
        self.start_ids = np.zeros(n_envs, np.int64)
        self.clear()

    @property
    def full(self):
        return self.size >= self.n_size

    def clear(self):                                                    # :221-230
        n, T = self.n_envs, self.n_size
        self.ptr, self.size = 0, 0
        self.observations = np.zeros((n, T) + self.obs_shape, self.obs_dtype)
        self.actions = np.zeros((n, T) + self.act_shape, np.float32)
        self.rewards = np.zeros((n, T), np.float32)
        self.returns = np.zeros((n, T), np.float32)
        self.values = np.zeros((n, T), np.float32)
        self.terminals = np.zeros((n, T), np.float32)
        self.advantages = np.zeros((n, T), np.float32)
        self.old_logp = np.zeros((n, T), np.float32)

    def store(self, obs, acts, rews, value, terminals, aux_info=None):  # :232-240
        p = self.ptr
        self.observations[:, p] = obs
        self.actions[:, p] = acts
        self.rewards[:, p] = rews
        self.values[:, p] = value
        self.terminals[:, p] = terminals
        if aux_info is not None:
            self.old_logp[:, p] = aux_info["old_logp"]
        self.ptr = (self.ptr + 1) % self.n_size
        self.size = min(self.size + 1, self.n_size)

    def finish_path(self, val, i):                                      # :242-265
        end = self.n_size if self.full else self.ptr
        sl = np.arange(self.start_ids[i], end)
        ret, adv = gae_finish_path(self.rewards[i, sl], self.values[i, sl], self.terminals[i, sl],
                                   val, self.gamma, self.gae_lam, self.use_gae)
        self.returns[i, sl] = ret
        self.advantages[i, sl] = adv
        self.start_ids[i] = self.ptr

    def sample(self, indexes):                                          # :267-287
        assert self.full, "Not enough transitions for on-policy buffer to random sample"
        env, step = np.divmod(np.asarray(indexes), self.n_size)
        adv = self.advantages[env, step]
        if self.use_advnorm:
            adv = (adv - np.mean(adv)) / (np.std(adv) + 1e-8)            # population std (ddof=0)
        return {
            "obs": self.observations[env, step], "actions": self.actions[env, step],
            "returns": self.returns[env, step], "values": self.values[env, step],
            "aux_batch": {"old_logp": self.old_logp[env, step]}, "batch_size": len(indexes),
            "advantages": adv.astype(np.float32),
        }


class OffPolicyBufferOracle:
    """Restatement of DummyOffPolicyBuffer(_Atari) (memory_tools.py:331-387, 601-630)."""

    def __init__(self, obs_shape, act_shape, n_envs, buffer_size, batch_size, obs_dtype=np.float32):
        assert buffer_size % n_envs == 0, "buffer_size must be divisible by the number of envs (parallels)"
        self.n_envs, self.n_size = n_envs, buffer_size // n_envs
        self.buffer_size, self.batch_size = buffer_size, batch_size
        n, S = self.n_envs, self.n_size
        self.ptr, self.size = 0, 0
        self.observations = np.zeros((n, S) + tuple(obs_shape), obs_dtype)
        self.next_observations = np.zeros((n, S) + tuple(obs_shape), obs_dtype)
        self.actions = np.zeros((n, S) + tuple(act_shape), np.float32)
        self.rewards = np.zeros((n, S), np.float32)
        self.terminals = np.zeros((n, S), np.float32)

    def store(self, obs, acts, rews, terminals, next_obs):              # :365-372
        p = self.ptr
        self.observations[:, p] = obs
        self.actions[:, p] = acts
        self.rewards[:, p] = rews
        self.terminals[:, p] = terminals
        self.next_observations[:, p] = next_obs
        self.ptr = (self.ptr + 1) % self.n_size
        self.size = min(self.size + 1, self.n_size)

    def sample_at(self, env, step):
        """The gather of ``sample`` (:379-386) with the random (env, step) choices made explicit."""
        return {"obs": self.observations[env, step], "actions": self.actions[env, step],
                "obs_next": self.next_observations[env, step], "rewards": self.rewards[env, step],
                "terminals": self.terminals[env, step], "batch_size": len(env)}

    def sample(self, batch_size=None, rng=np.random):                   # :374-387
        bs = self.batch_size if batch_size is None else batch_size
        env = rng.choice(self.n_envs, bs)
        step = rng.choice(self.size, bs)
        return self.sample_at(env, step)


# --------------------------------------------------------------------------------------
# Running mean / std and obs / reward processing   statistic_tools.py:65-185, agent.py:262-294
# --------------------------------------------------------------------------------------
class RunningMeanStdOracle:
    def __init__(self, shape, epsilon=1e-4):
        self.mean = np.zeros(shape, np.float32)
        self.var = np.ones(shape, np.float32)
        self.count = epsilon

    @property
    def std(self):
        return np.sqrt(self.var)

    def update(self, x):                                                # :117-147
        x = np.asarray(x)
        self.update_from_moments(np.mean(x, axis=0), np.square(np.std(x, axis=0)), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):  # :149-185
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        M2 = m_a + m_b + np.square(delta) * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, M2 / tot, tot


def process_observation(obs, rms, obsnorm_range=5.0):                  # agent.py:262-283
    return np.clip((obs - rms.mean) / (rms.std + EPS), -obsnorm_range, obsnorm_range)


def process_reward(rew, ret_rms, rewnorm_range=5.0):                   # agent.py:285-294
    std = np.clip(ret_rms.std, 0.1, 100)
    return np.clip(rew / std, -rewnorm_range, rewnorm_range)


# --------------------------------------------------------------------------------------
# Dense layers with explicit backward                          layers.py:16-33 (mlp_block)
# --------------------------------------------------------------------------------------
def act_fwd(z, kind):
    if kind is None or kind == "none":
        return z
    if kind == "leaky_relu":
        return np.where(z > 0, z, z * z.dtype.type(0.01))
    if kind == "relu":
        return np.maximum(z, 0)
    if kind == "tanh":
        return np.tanh(z)
    if kind == "sigmoid":
        return 1 / (1 + np.exp(-z))
    raise ValueError(kind)


def act_bwd_from_out(y, kind):
    """d act / d z expressed with the activation OUTPUT y (what torch's backward kernels use)."""
    if kind is None or kind == "none":
        return np.ones_like(y)
    if kind == "leaky_relu":
        return np.where(y > 0, y.dtype.type(1), y.dtype.type(0.01))
    if kind == "relu":
        return (y > 0).astype(y.dtype)
    if kind == "tanh":
        return 1 - y * y
    if kind == "sigmoid":
        return y * (1 - y)
    raise ValueError(kind)


class MLP:
    """A stack of Linear(+activation) layers held as (W [out,in], b [out], act) -- nn.Sequential of mlp_blocks."""

    def __init__(self, layers):
        self.layers = layers          # list of dict(W=..., b=..., act=...)

    def forward(self, x):
        self.cache = [x]
        for L in self.layers:
            x = act_fwd(x @ L["W"].T + L["b"], L["act"])
            self.cache.append(x)
        return x

    def backward(self, dy, need_dx=True):
        grads = []
        for li in reversed(range(len(self.layers))):
            L = self.layers[li]
            y, xin = self.cache[li + 1], self.cache[li]
            dz = dy * act_bwd_from_out(y, L["act"])
            grads.append((dz.T @ xin, dz.sum(0)))
            dy = dz @ L["W"] if (li > 0 or need_dx) else None
        grads.reverse()
        return dy, grads


def log_softmax(z):
    m = z.max(-1, keepdims=True)
    s = z - m
    return s - np.log(np.exp(s).sum(-1, keepdims=True))


# --------------------------------------------------------------------------------------
# Optimiser: clip_grad_norm_ + Adam(eps=1e-5) + LinearLR   (PyTorch, restated)
# --------------------------------------------------------------------------------------
class AdamOracle:
    """torch.optim.Adam (betas .9/.999, no amsgrad) over a dict name->array; LinearLR(start=1, end=ef, total)."""

    def __init__(self, params, lr, eps=1e-5, weight_decay=0.0, end_factor=1.0, total_iters=1):
        self.params = params
        self.base_lr, self.eps, self.wd = lr, eps, weight_decay
        self.end_factor, self.total_iters = end_factor, max(int(total_iters), 1)
        self.t = 0            # optimiser steps taken
        self.sched_steps = 0  # scheduler steps taken
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    @property
    def lr(self):
        k = min(self.sched_steps, self.total_iters)
        return self.base_lr * (1.0 + (self.end_factor - 1.0) * k / self.total_iters)

    @staticmethod
    def clip_grad_norm_(grads, max_norm):
        tot = math.sqrt(sum(float(np.sum(np.square(g.astype(np.float64)))) for g in grads.values()))
        coef = min(max_norm / (tot + 1e-6), 1.0)
        for k in grads:
            grads[k] = (grads[k] * grads[k].dtype.type(coef))
        return tot

    def step(self, grads):
        self.t += 1
        b1, b2 = 0.9, 0.999
        lr = self.lr
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                continue           # parameters without grad (frozen targets) are skipped by torch
            T = p.dtype.type
            if self.wd != 0:
                g = g + T(self.wd) * p
            self.m[k] = self.m[k] + (g - self.m[k]) * T(1 - b1)
            self.v[k] = self.v[k] * T(b2) + T(1 - b2) * g * g
            denom = np.sqrt(self.v[k]) / T(bc2_sqrt) + T(self.eps)
            p -= T(step_size) * (self.m[k] / denom)
        self.sched_steps += 1      # scheduler.step() after optimizer.step()


# --------------------------------------------------------------------------------------
# PPO-clip                                                        ppo_learner.py:35-95
# --------------------------------------------------------------------------------------
def build_actor_critic_layers(sd, prefix_rep="representation.model", actor_key="actor.logits",
                              critic_key="critic.values", act="leaky_relu", activation_action=None):
    """Group a reference-style state_dict into (rep, actor, critic) layer lists."""
    def collect(prefix, last_act):
        idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in sd if k.startswith(prefix + ".")})
        layers = [dict(W=sd[f"{prefix}.{i}.weight"], b=sd[f"{prefix}.{i}.bias"], act=act, name=f"{prefix}.{i}")
                  for i in idx]
        if layers and last_act != "keep":
            layers[-1]["act"] = last_act
        return layers
    rep = collect(prefix_rep, "keep")
    actor = collect(actor_key, activation_action)
    critic = collect(critic_key, None)
    return rep, actor, critic


def actor_critic_forward(sd, obs, dist="categorical", act="leaky_relu", activation_action=None):
    """SharedActorCritic.forward (actor_critic.py:51-59): returns (actor head output, values)."""
    actor_key = "actor.logits" if dist == "categorical" else "actor.mu"
    rep_l, actor_l, critic_l = build_actor_critic_layers(sd, actor_key=actor_key, act=act,
                                                         activation_action=activation_action)
    obs = np.asarray(obs, np.float32)
    h = MLP(rep_l).forward(obs) if rep_l else obs
    return MLP(actor_l).forward(h), MLP(critic_l).forward(h)[:, 0]


def ppo_forward_backward(sd, batch, cfg, dist="categorical", act="leaky_relu", activation_action=None, loss_kind="ppo"):
    """Forward + loss + backward of PPO_Learner.update (ppo_learner.py:46-62); with loss_kind="a2c" of
    A2C_Learner.update (a2c_learner.py:41-53): a_loss = -(adv * log_prob).mean(), no ratio / old_logp.

    sd: dict name->np.ndarray in the reference's state_dict naming.
    batch: obs, actions, returns, advantages, old_logp.
    Returns (info dict of intermediates, grads dict name->array).
    """
    actor_key = "actor.logits" if dist == "categorical" else "actor.mu"
    rep_l, actor_l, critic_l = build_actor_critic_layers(sd, actor_key=actor_key, act=act,
                                                         activation_action=activation_action)
    dt = batch["obs"].dtype if batch["obs"].dtype in (np.float32, np.float64) else np.float32
    dt = np.dtype(dt).type
    rep, actor, critic = MLP(rep_l), MLP(actor_l), MLP(critic_l)
    obs = batch["obs"].astype(dt)
    h = rep.forward(obs) if rep_l else obs
    out_a = actor.forward(h)
    v = critic.forward(h)[:, 0]
    B = obs.shape[0]
    adv, ret = batch["advantages"].astype(dt), batch["returns"].astype(dt)
    a2c, ppokl = loss_kind == "a2c", loss_kind == "ppokl"
    old_logp = np.zeros(B, dt) if (a2c or ppokl) else batch["old_logp"].astype(dt)
    clip = dt(cfg.get("clip_range", 0.0))

    if dist == "categorical":
        a = batch["actions"].astype(np.int64)
        lsm = log_softmax(out_a)
        p = np.exp(lsm)
        logp = lsm[np.arange(B), a]                                     # Categorical.log_prob
        ent = -(p * lsm).sum(-1)                                        # Categorical.entropy
        if ppokl:                                                       # ppokl_learner.py:51-54 (old_dists: stored logits)
            q_old = log_softmax(batch["old_a"].astype(dt))              # Categorical(logits=...) normalises
            old_logp = q_old[np.arange(B), a]
            kl_row = (p * (lsm - q_old)).sum(-1)                        # torch kl_divergence(Categorical, Categorical)
            kl = kl_row.mean()
    else:
        log_std = sd["actor.log_std"].astype(dt)
        std = np.exp(log_std)
        x = batch["actions"].astype(dt)
        var = std * std
        # Normal.log_prob: -((x-mu)^2)/(2 var) - log(std) - log(sqrt(2 pi))   (distributions.py:179-180)
        lp = -((x - out_a) ** 2) / (2 * var) - log_std - dt(math.log(math.sqrt(2 * math.pi)))
        logp = lp.sum(-1)
        ent = np.broadcast_to((dt(0.5 + 0.5 * math.log(2 * math.pi)) + log_std).sum(-1), (B,)).astype(dt)
        if ppokl:
            mu_o, std_o = batch["old_a"].astype(dt), batch["old_b"].astype(dt)
            old_logp = (-((x - mu_o) ** 2) / (2 * std_o * std_o) - np.log(std_o) - dt(math.log(math.sqrt(2 * math.pi)))).sum(-1)
            var_ratio = (std / std_o) ** 2                             # torch kl_divergence(Normal, Normal): ELEMENTWISE,
            t1 = ((out_a - mu_o) / std_o) ** 2                          # so .mean() below averages over B x A
            kl_el = dt(0.5) * (var_ratio + t1 - 1 - np.log(var_ratio))
            kl = kl_el.mean()

    ratio = np.exp(logp - old_logp)                                     # :52
    s1 = np.clip(ratio, 1 - clip, 1 + clip) * adv                       # :53
    s2 = adv * ratio                                                    # :54
    a_loss = -np.minimum(s1, s2).mean()                                 # :55
    if a2c:
        a_loss = -(adv * logp).mean()                                   # a2c_learner.py:47
    if ppokl:
        kl_coef = dt(cfg["kl_coef"])
        a_loss = -(ratio * adv).mean() + kl_coef * kl                   # ppokl_learner.py:58
    c_loss = ((v - ret) ** 2).mean()                                    # :57
    e_loss = ent.mean()                                                 # :59
    loss = a_loss - dt(cfg["ent_coef"]) * e_loss + dt(cfg["vf_coef"]) * c_loss   # :60

    # ---- backward ----
    invB = dt(1.0 / B)
    # d(-mean(min(s1,s2)))/d ratio: torch.minimum routes the gradient to s1 when s1 < s2, to s2 when
    # s2 < s1 and splits it 0.5/0.5 on ties; clamp passes gradient only strictly inside... (clamp
    # backward mask is  lo <= x <= hi).
    inside = ((ratio >= 1 - clip) & (ratio <= 1 + clip)).astype(dt)
    w1 = np.where(s1 < s2, dt(1), np.where(s1 == s2, dt(0.5), dt(0)))
    w2 = dt(1) - w1
    dratio = -(w1 * inside * adv + w2 * adv) * invB
    dlogp = dratio * ratio
    if a2c:
        dlogp = -adv * invB
    if ppokl:
        dlogp = -adv * ratio * invB
    dv = dt(cfg["vf_coef"]) * 2 * (v - ret) * invB
    grads = {}
    if dist == "categorical":
        onehot = np.zeros_like(out_a)
        onehot[np.arange(B), a] = 1
        dlogits = dlogp[:, None] * (onehot - p)
        # d entropy / d logits_j = -p_j (lsm_j + H)
        dent = -p * (lsm + ent[:, None])
        dlogits = dlogits + (-dt(cfg["ent_coef"]) * invB) * dent
        if ppokl:                                                       # d kl_row / d logits_j = p_j (l_j - q_j - kl_row)
            dlogits = dlogits + (kl_coef * invB) * p * (lsm - q_old - kl_row[:, None])
        dout_a = dlogits
    else:
        dmu = dlogp[:, None] * (x - out_a) / var
        dlog_std = (dlogp[:, None] * (((x - out_a) ** 2) / var - 1)).sum(0)
        dlog_std = dlog_std + (-dt(cfg["ent_coef"])) * np.ones_like(log_std)   # d mean(ent)/d log_std = 1
        if ppokl:
            w = kl_coef / dt(B * out_a.shape[1])
            dmu = dmu + w * (out_a - mu_o) / (std_o * std_o)
            dlog_std = dlog_std + w * (var_ratio - 1).sum(0)
        grads["actor.log_std"] = dlog_std.astype(dt)
        dout_a = dmu
    dh_a, g_actor = actor.backward(dout_a, need_dx=bool(rep_l))
    dh_c, g_critic = critic.backward(dv[:, None], need_dx=bool(rep_l))
    for L, (gw, gb) in zip(actor_l, g_actor):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    for L, (gw, gb) in zip(critic_l, g_critic):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    if rep_l:
        _, g_rep = rep.backward(dh_a + dh_c, need_dx=False)
        for L, (gw, gb) in zip(rep_l, g_rep):
            grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb

    cr = ((ratio < 1 - clip).sum() + (ratio > 1 + clip).sum()) / ratio.shape[0]   # :70
    info = dict(logits_or_mu=out_a, v_pred=v, log_prob=logp, ratio=ratio, surrogate1=s1, surrogate2=s2,
                a_loss=a_loss, c_loss=c_loss, e_loss=e_loss, loss=loss, clip_ratio=cr,
                predict_value=v.mean(), entropy=ent)
    if ppokl:
        info["kl"] = kl
    return info, grads


def ppokl_adapt(kl_coef, kl, target_kl):
    """The coefficient schedule of ppokl_learner.py:62-66 (a Python float; kl is a float32 tensor there)."""
    if np.float32(kl) > np.float32(target_kl * 1.5):
        kl_coef = kl_coef * 2.0
    elif np.float32(kl) < np.float32(target_kl * 0.5):
        kl_coef = kl_coef / 2.0
    return float(np.clip(kl_coef, 0.1, 20))


def pg_forward_backward(sd, batch, cfg, dist="categorical", act="leaky_relu", activation_action=None):
    """PG_Learner.update (pg_learner.py:30-71) on VanillaPolicyGradient(actor): loss = -(returns * log_prob).mean()
    - ent_coef * entropy.mean().  sd names: actor.representation.model.*, actor.actor_head.{logits|mu}.*, actor.actor_head.log_std."""
    dt = np.float32
    key = "actor.actor_head.logits" if dist == "categorical" else "actor.actor_head.mu"
    rep_l = collect_seq(sd, "actor.representation.model", act, last_act=act)
    act_l = collect_seq(sd, key, act, last_act=activation_action)
    rep, actor = MLP(rep_l), MLP(act_l)
    obs = batch["obs"].astype(dt)
    B = obs.shape[0]
    h = rep.forward(obs) if rep_l else obs
    out = actor.forward(h)
    ret = batch["returns"].astype(dt)
    invB = dt(1.0 / B)
    grads = {}
    if dist == "categorical":
        a = batch["actions"].astype(np.int64)
        lsm = log_softmax(out)
        p = np.exp(lsm)
        logp = lsm[np.arange(B), a]
        ent = -(p * lsm).sum(-1)
        onehot = np.zeros_like(out); onehot[np.arange(B), a] = 1
        dout = (-ret * invB)[:, None] * (onehot - p) + (-dt(cfg["ent_coef"]) * invB) * (-p * (lsm + ent[:, None]))
    else:
        log_std = sd["actor.actor_head.log_std"].astype(dt)
        var = np.exp(log_std) ** 2
        x = batch["actions"].astype(dt)
        logp = (-((x - out) ** 2) / (2 * var) - log_std - dt(math.log(math.sqrt(2 * math.pi)))).sum(-1)
        ent = np.broadcast_to((dt(0.5 + 0.5 * math.log(2 * math.pi)) + log_std).sum(-1), (B,)).astype(dt)
        dlogp = -ret * invB
        dout = dlogp[:, None] * (x - out) / var
        grads["actor.actor_head.log_std"] = ((dlogp[:, None] * (((x - out) ** 2) / var - 1)).sum(0)
                                             - dt(cfg["ent_coef"]) * np.ones_like(log_std)).astype(dt)
    a_loss = -(ret * logp).mean()
    e_loss = ent.mean()
    loss = a_loss - dt(cfg["ent_coef"]) * e_loss
    dh, g_a = actor.backward(dout, need_dx=bool(rep_l))
    for L, (gw, gb) in zip(act_l, g_a):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    if rep_l:
        _, g_rep = rep.backward(dh, need_dx=False)
        for L, (gw, gb) in zip(rep_l, g_rep):
            grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    return dict(log_prob=logp, a_loss=a_loss, e_loss=e_loss, loss=loss, out=out), grads


def ppo_update(sd, opt, batch, cfg, **kw):
    """One full PPO_Learner.update: fwd/bwd, clip_grad_norm_, Adam, LinearLR (ppo_learner.py:35-95)."""
    info, grads = ppo_forward_backward(sd, batch, cfg, **kw)
    raw = {k: g.copy() for k, g in grads.items()}
    if cfg.get("use_grad_clip", True):
        info["grad_norm"] = AdamOracle.clip_grad_norm_(grads, cfg["grad_clip_norm"])
    info["clipped_grads"] = {k: g.copy() for k, g in grads.items()}     # what Adam consumes (tests propagate tolerances through it)
    opt.step(grads)
    info["learning_rate"] = opt.lr
    return info, raw


# --------------------------------------------------------------------------------------
# DQN                                                              dqn_learner.py:28-75
# --------------------------------------------------------------------------------------
def collect_seq(sd, prefix, act, last_act=None):
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in sd if k.startswith(prefix + ".")})
    layers = [dict(W=sd[f"{prefix}.{i}.weight"], b=sd[f"{prefix}.{i}.bias"], act=act, name=f"{prefix}.{i}")
              for i in idx]
    if layers:
        layers[-1]["act"] = last_act
    return layers


def td_loss_mean(pred, target, huber_delta=0.0):
    """nn.MSELoss()(pred, target) (dqn_learner.py:25,46) and its gradient w.r.t. pred; with huber_delta > 0 nn.HuberLoss(delta,
    reduction "mean") instead -- the loss the reference's learners build behind `use_huber_loss` (marl_learner.py:193-197):
    0.5 z^2 where |z| < delta, delta (|z| - 0.5 delta) beyond (torch's definition)."""
    z = pred - target
    B = z.dtype.type(z.size)
    if not huber_delta or huber_delta <= 0:
        return (z ** 2).mean(), 2 * z / B
    d = z.dtype.type(huber_delta)
    az = np.abs(z)
    quad = az < d
    loss = np.where(quad, z.dtype.type(0.5) * z * z, d * (az - z.dtype.type(0.5) * d)).mean()
    return loss, np.where(quad, z, d * np.sign(z)) / B


def dqn_forward_backward(sd, batch, cfg, act="relu"):
    """DQN_Learner.update forward/loss/backward for an MLP-representation DeepQNetwork.

    State-dict names follow deep_q_network.py:19-60: representation.model.*, eval_Q_head.q_value.*,
    target_representation.model.*, target_Q_head.q_value.*.
    """
    dt = np.float32 if cfg.get("dtype", "f32") == "f32" else np.float64
    rep_l = collect_seq(sd, "representation.model", act, last_act=act)
    q_l = collect_seq(sd, "eval_Q_head.q_value", act)
    trep_l = collect_seq(sd, "target_representation.model", act, last_act=act)
    tq_l = collect_seq(sd, "target_Q_head.q_value", act)
    rep, q, trep, tq = MLP(rep_l), MLP(q_l), MLP(trep_l), MLP(tq_l)
    obs, nxt = batch["obs"].astype(dt), batch["obs_next"].astype(dt)
    B = obs.shape[0]
    h = rep.forward(obs) if rep_l else obs
    evalQ = q.forward(h)                                                # :39
    targetQ_all = tq.forward(trep.forward(nxt) if trep_l else nxt)      # :40
    a = batch["actions"].astype(np.int64)
    predictQ = evalQ[np.arange(B), a]                                   # :42
    if cfg.get("double_q", False):                                      # DDQN_Learner (ddqn_learner.py:39-47): the target
        h2 = rep.forward(nxt) if rep_l else nxt                         # action is the eval network's argmax on next_obs
        tmax = targetQ_all[np.arange(B), q.forward(h2).argmax(-1)]
        h = rep.forward(obs) if rep_l else obs                          # (restore the activations backward() uses)
        evalQ = q.forward(h)
    else:
        tmax = targetQ_all.max(-1)                                      # :43
    g = dt(cfg["gamma"])
    targetQ = batch["rewards"].astype(dt) + g * (1 - batch["terminals"].astype(dt)) * tmax   # :44
    loss, dpred = td_loss_mean(predictQ, targetQ, cfg.get("huber_delta", 0.0))   # :46
    dQ = np.zeros_like(evalQ)
    dQ[np.arange(B), a] = dpred
    dh, g_q = q.backward(dQ, need_dx=bool(rep_l))
    grads = {}
    for L, (gw, gb) in zip(q_l, g_q):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    if rep_l:
        _, g_rep = rep.backward(dh, need_dx=False)
        for L, (gw, gb) in zip(rep_l, g_rep):
            grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    info = dict(evalQ=evalQ, predictQ=predictQ, targetQ=targetQ, loss=loss, dQ=dQ)
    return info, grads


def dueldqn_forward_backward(sd, batch, cfg, act="relu"):
    """DuelDQN_Learner.update (dueldqn_learner.py:28-75) on DuelingDeepQNetwork: DuelingQValueHead (q_head.py:42-80) with
    v_model / a_model streams, Q = V + (A - mean(A)); the update rule is DQN's."""
    dt = np.float32
    rep_l = collect_seq(sd, "representation.model", act, last_act=act)
    trep_l = collect_seq(sd, "target_representation.model", act, last_act=act)
    nets = {k: MLP(collect_seq(sd, k, act)) for k in ("eval_Q_head.v_model", "eval_Q_head.a_model", "target_Q_head.v_model",
                                                      "target_Q_head.a_model")}
    rep, trep = MLP(rep_l), MLP(trep_l)
    obs, nxt = batch["obs"].astype(dt), batch["obs_next"].astype(dt)
    B = obs.shape[0]
    h = rep.forward(obs) if rep_l else obs
    V, Adv = nets["eval_Q_head.v_model"].forward(h), nets["eval_Q_head.a_model"].forward(h)
    evalQ = V + (Adv - Adv.mean(-1, keepdims=True))                    # q_head.py:75-77
    ht = trep.forward(nxt) if trep_l else nxt
    Vt, At = nets["target_Q_head.v_model"].forward(ht), nets["target_Q_head.a_model"].forward(ht)
    tmax = (Vt + (At - At.mean(-1, keepdims=True))).max(-1)             # dueldqn_learner.py:43
    a = batch["actions"].astype(np.int64)
    predictQ = evalQ[np.arange(B), a]                                   # :42
    targetQ = batch["rewards"].astype(dt) + dt(cfg["gamma"]) * (1 - batch["terminals"].astype(dt)) * tmax   # :44
    loss = ((predictQ - targetQ) ** 2).mean()                           # :46
    dQ = np.zeros_like(evalQ)
    dQ[np.arange(B), a] = 2 * (predictQ - targetQ) / dt(B)
    dV = dQ.sum(-1, keepdims=True)
    dA = dQ - dQ.sum(-1, keepdims=True) / dt(Adv.shape[1])
    dh_v, g_v = nets["eval_Q_head.v_model"].backward(dV, need_dx=bool(rep_l))
    dh_a, g_a = nets["eval_Q_head.a_model"].backward(dA, need_dx=bool(rep_l))
    grads = {}
    for key, gl in (("eval_Q_head.v_model", g_v), ("eval_Q_head.a_model", g_a)):
        for L, (gw, gb) in zip(nets[key].layers, gl):
            grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    if rep_l:
        _, g_rep = rep.backward(dh_v + dh_a, need_dx=False)
        for L, (gw, gb) in zip(rep_l, g_rep):
            grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    return dict(evalQ=evalQ, predictQ=predictQ, targetQ=targetQ, loss=loss, dQ=dQ), grads


def dqn_copy_target(sd):                                                # deep_q_network.py:95-99
    for k in list(sd):
        if k.startswith("representation."):
            sd["target_" + k][...] = sd[k]
        elif k.startswith("eval_Q_head."):
            sd["target_Q_head." + k[len("eval_Q_head."):]][...] = sd[k]


def egreedy_select(greedy, random_actions, uniforms, eps):             # off_policy.py:138-141
    return np.where(uniforms < eps, random_actions, greedy)


def egreedy_schedule(start, end, decay_steps, n_envs, n_vector_steps):  # off_policy.py:119-127, dqn_agent.py:28-30
    delta = (start - end) / (decay_steps / n_envs)
    e, cur, out = start, 0, []
    for _ in range(n_vector_steps):
        out.append(e)
        cur += n_envs
        if e > end:
            e = start - cur * delta
    return out


# --------------------------------------------------------------------------------------
# QMIX (feed-forward)      qmix_learner.py:24-112, iql_learner.py:37-83, q_mix_head.py:66-95
# --------------------------------------------------------------------------------------
def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def qmix_mixer_forward(sd, prefix, agent_qs, states):
    """QMIX_Mixer.forward (q_mix_head.py:66-95). agent_qs [R,N], states [R,S]. Returns q_tot [R] + cache."""
    dt = agent_qs.dtype.type
    W = lambda n: sd[f"{prefix}.{n}"]
    N = agent_qs.shape[1]
    a1 = np.maximum(states @ W("hyper_w_1.0.weight").T + W("hyper_w_1.0.bias"), 0)
    w1_raw = a1 @ W("hyper_w_1.2.weight").T + W("hyper_w_1.2.bias")
    H = w1_raw.shape[1] // N
    w1 = np.abs(w1_raw).reshape(-1, N, H)
    b1 = states @ W("hyper_b_1.weight").T + W("hyper_b_1.bias")
    pre = np.einsum("rn,rnh->rh", agent_qs, w1) + b1
    hidden = elu(pre)
    a2 = np.maximum(states @ W("hyper_w_2.0.weight").T + W("hyper_w_2.0.bias"), 0)
    w2_raw = a2 @ W("hyper_w_2.2.weight").T + W("hyper_w_2.2.bias")
    w2 = np.abs(w2_raw)
    a3 = np.maximum(states @ W("hyper_b_2.0.weight").T + W("hyper_b_2.0.bias"), 0)
    b2 = (a3 @ W("hyper_b_2.2.weight").T + W("hyper_b_2.2.bias"))[:, 0]
    q_tot = (hidden * w2).sum(-1) + b2
    cache = dict(states=states, agent_qs=agent_qs, a1=a1, w1_raw=w1_raw, w1=w1, pre=pre, hidden=hidden,
                 a2=a2, w2_raw=w2_raw, w2=w2, a3=a3, N=N, H=H)
    return q_tot.astype(dt), cache


def qmix_mixer_backward(sd, prefix, cache, dq_tot):
    """Backward of the mixer: returns (d agent_qs [R,N], grads dict)."""
    W = lambda n: sd[f"{prefix}.{n}"]
    c = cache
    s = c["states"]
    g = {}
    dhidden = dq_tot[:, None] * c["w2"]
    dw2 = dq_tot[:, None] * c["hidden"]
    db2 = dq_tot
    dpre = dhidden * np.where(c["pre"] > 0, 1, np.exp(np.minimum(c["pre"], 0)))
    dq = np.einsum("rh,rnh->rn", dpre, c["w1"])
    dw1 = np.einsum("rn,rh->rnh", c["agent_qs"], dpre).reshape(len(s), -1)
    db1 = dpre
    # hyper_b_1 (single Linear)
    g[f"{prefix}.hyper_b_1.weight"] = db1.T @ s
    g[f"{prefix}.hyper_b_1.bias"] = db1.sum(0)
    # hyper_w_1: Linear-ReLU-Linear then abs
    dw1_raw = dw1 * np.sign(c["w1_raw"])
    g[f"{prefix}.hyper_w_1.2.weight"] = dw1_raw.T @ c["a1"]
    g[f"{prefix}.hyper_w_1.2.bias"] = dw1_raw.sum(0)
    da1 = (dw1_raw @ W("hyper_w_1.2.weight")) * (c["a1"] > 0)
    g[f"{prefix}.hyper_w_1.0.weight"] = da1.T @ s
    g[f"{prefix}.hyper_w_1.0.bias"] = da1.sum(0)
    # hyper_w_2
    dw2_raw = dw2 * np.sign(c["w2_raw"])
    g[f"{prefix}.hyper_w_2.2.weight"] = dw2_raw.T @ c["a2"]
    g[f"{prefix}.hyper_w_2.2.bias"] = dw2_raw.sum(0)
    da2 = (dw2_raw @ W("hyper_w_2.2.weight")) * (c["a2"] > 0)
    g[f"{prefix}.hyper_w_2.0.weight"] = da2.T @ s
    g[f"{prefix}.hyper_w_2.0.bias"] = da2.sum(0)
    # hyper_b_2
    db2c = db2[:, None]
    g[f"{prefix}.hyper_b_2.2.weight"] = db2c.T @ c["a3"]
    g[f"{prefix}.hyper_b_2.2.bias"] = db2c.sum(0)
    da3 = (db2c @ W("hyper_b_2.2.weight")) * (c["a3"] > 0)
    g[f"{prefix}.hyper_b_2.0.weight"] = da3.T @ s
    g[f"{prefix}.hyper_b_2.0.bias"] = da3.sum(0)
    return dq, g


def qmix_forward_backward(sd, batch, cfg, act="relu", group="shared"):
    """QMIX_Learner.update (feed-forward branch) with parameter sharing (one group).

    sd names follow value_factorization.py:17-52:
      individual_q_networks.<group>.representation.obs_representation.model.<i>.{weight,bias}
      individual_q_networks.<group>.critic_head.q_value.<i>.{weight,bias}
      target_individual_q_networks.<group>....   eval_Qtot.* / target_Qtot.*
    batch (already stacked like build_training_data, marl_learner.py:319-408):
      obs [B,N,O], obs_next [B,N,O], actions [B,N] (float), rewards [B,N], terminals [B,N] (bool/float),
      agent_mask [B,N], avail_actions [B,N,A], avail_actions_next [B,N,A], state [B,S], state_next [B,S]
    """
    dt = np.float32
    pe = f"individual_q_networks.{group}"
    pt = f"target_individual_q_networks.{group}"
    rep_l = collect_seq(sd, f"{pe}.representation.obs_representation.model", act, last_act=act)
    q_l = collect_seq(sd, f"{pe}.critic_head.q_value", act)
    trep_l = collect_seq(sd, f"{pt}.representation.obs_representation.model", act, last_act=act)
    tq_l = collect_seq(sd, f"{pt}.critic_head.q_value", act)
    B, N, O = batch["obs"].shape
    obs = batch["obs"].reshape(B * N, O).astype(dt)
    nxt = batch["obs_next"].reshape(B * N, O).astype(dt)
    rep, q = MLP(rep_l), MLP(q_l)
    q_eval = q.forward(rep.forward(obs))                                 # iql_learner.py:41-47
    A = q_eval.shape[-1]
    q_next = MLP(tq_l).forward(MLP(trep_l).forward(nxt)).copy()         # :63-66
    use_mask = cfg.get("use_actions_mask", True)
    if cfg.get("double_q", True):                                       # :68-71
        qn_eval = MLP(q_l).forward(MLP(rep_l).forward(nxt)).copy()
        if use_mask:
            qn_eval[batch["avail_actions_next"].reshape(B * N, A) == 0] = -1e10   # value_factorization.py:87-90
        a_next = qn_eval.argmax(-1)
    if use_mask:
        q_next[batch["avail_actions_next"].reshape(B * N, A) == 0] = -1e10      # iql_learner.py:75-81
    rewards_tot = batch["rewards"].astype(dt).mean(1)                   # qmix_learner.py:34
    terminals_tot = batch["terminals"].astype(bool).all(1).astype(dt)   # :35
    mask = batch["agent_mask"].astype(dt)                               # valid_mask, :45
    a_taken = batch["actions"].reshape(B * N).astype(np.int64)
    rows = np.arange(B * N)
    q_eval_taken = q_eval[rows, a_taken].reshape(B, N)                  # :48-50
    if cfg.get("double_q", True):
        q_next_taken = q_next[rows, a_next].reshape(B, N)               # :52-55
    else:
        q_next_taken = q_next.max(-1).reshape(B, N)                     # :57-58
    mixer = cfg.get("mixer", "qmix")
    if mixer == "iql":                                                  # iql_learner.py:98-117: per-agent TD, no mixing
        r_a, d_a = batch["rewards"].astype(dt), batch["terminals"].astype(dt)
        target = r_a + (1 - d_a) * dt(cfg["gamma"]) * q_next_taken                  # :113
        td = (q_eval_taken - target) * mask                                        # :116
        loss = (td ** 2).sum() / mask.sum()                                         # :117
        dq_taken = (2 * td * mask / mask.sum()).astype(dt).reshape(B * N)
        grads = {}
        q_tot_eval, q_tot_next, pred = q_eval_taken.reshape(-1), q_next_taken.reshape(-1), q_eval_taken.mean()
    else:
        q_eval_m = q_eval_taken * mask                                  # :60
        q_next_m = q_next_taken * mask                                  # :61
        if mixer == "vdn":                                              # VDN_Mixer: sum over agents (vdn_learner.py:74-75)
            q_tot_eval, q_tot_next = q_eval_m.sum(1), q_next_m.sum(1)
        else:
            q_tot_eval, cache = qmix_mixer_forward(sd, "eval_Qtot", q_eval_m, batch["state"].astype(dt))
            q_tot_next, _ = qmix_mixer_forward(sd, "target_Qtot", q_next_m, batch["state_next"].astype(dt))
        target = rewards_tot + (1 - terminals_tot) * dt(cfg["gamma"]) * q_tot_next     # :78
        loss = ((q_tot_eval - target) ** 2).mean()                      # :86
        dq_tot = (2 * (q_tot_eval - target) / dt(B)).astype(dt)
        if mixer == "vdn":
            dq_m, grads = np.repeat(dq_tot[:, None], N, 1), {}
        else:
            dq_m, grads = qmix_mixer_backward(sd, "eval_Qtot", cache, dq_tot)
        dq_taken = (dq_m * mask).reshape(B * N)
        pred = q_tot_eval.mean()
    dQ = np.zeros_like(q_eval)
    dQ[rows, a_taken] = dq_taken
    dh, g_q = q.backward(dQ, need_dx=True)
    for L, (gw, gb) in zip(q_l, g_q):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    _, g_rep = rep.backward(dh, need_dx=False)
    for L, (gw, gb) in zip(rep_l, g_rep):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    info = dict(q_eval=q_eval.reshape(B, N, A), q_next=q_next.reshape(B, N, A), q_tot_eval=q_tot_eval,
                q_tot_next=q_tot_next, q_tot_target=target, loss=loss, predictQ=pred,
                rewards_tot=rewards_tot, terminals_tot=terminals_tot)
    if cfg.get("double_q", True):
        info["actions_next"] = a_next.reshape(B, N)
    return info, grads


class PerBufferOracle:
    """PerOffPolicyBuffer's priority machinery (memory_tools.py:471-598; SumSegmentTree / MinSegmentTree,
    segtree_tool.py:24-230) with float64 trees [n_envs][2*capacity]; the transition arrays themselves are
    OffPolicyBufferOracle's."""

    def __init__(self, n_envs, n_size, batch_size, alpha):
        self.n_envs, self.n_size, self.k, self.alpha = n_envs, n_size, batch_size // n_envs, float(alpha)
        cap = 1
        while cap < n_size:
            cap *= 2
        self.cap = cap
        self.sum = np.zeros((n_envs, 2 * cap), np.float64)
        self.min = np.full((n_envs, 2 * cap), np.inf, np.float64)
        self.max_priority = np.ones(n_envs, np.float64)
        self.ptr = self.size = 0

    def _set(self, e, idx, val):                               # segtree_tool.py:98-113
        i = idx + self.cap
        self.sum[e, i] = self.min[e, i] = val
        i //= 2
        while i >= 1:
            self.sum[e, i] = self.sum[e, 2 * i] + self.sum[e, 2 * i + 1]
            self.min[e, i] = min(self.min[e, 2 * i], self.min[e, 2 * i + 1])
            i //= 2

    def _reduce(self, e, start, end, node, ns, ne):            # :41-63
        if start == ns and end == ne:
            return self.sum[e, node]
        mid = (ns + ne) // 2
        if end <= mid:
            return self._reduce(e, start, end, 2 * node, ns, mid)
        if mid + 1 <= start:
            return self._reduce(e, start, end, 2 * node + 1, mid + 1, ne)
        return self._reduce(e, start, mid, 2 * node, ns, mid) + self._reduce(e, mid + 1, end, 2 * node + 1, mid + 1, ne)

    def store(self):                                           # memory_tools.py:536-541
        for e in range(self.n_envs):
            self._set(e, self.ptr, self.max_priority[e] ** self.alpha)
        self.ptr = (self.ptr + 1) % self.n_size
        self.size = min(self.size + 1, self.n_size)

    def sample(self, beta, uniforms):                          # :499-507, 542-565
        steps = np.zeros((self.n_envs, self.k), np.int64)
        weights = np.zeros((self.n_envs, self.k), np.float64)
        for e in range(self.n_envs):
            p_total = self._reduce(e, 0, self.size - 2, 1, 0, self.cap - 1)        # sum(0, size - 1), end exclusive
            every = p_total / self.k
            p_min = self.min[e, 1] / self.sum[e, 1]
            max_weight = p_min * self.size ** (-beta)
            for i in range(self.k):
                mass = uniforms[e, i] * every + i * every
                idx = 1
                while idx < self.cap:                          # find_prefixsum_idx (segtree_tool.py:160-170)
                    if self.sum[e, 2 * idx] > mass:
                        idx = 2 * idx
                    else:
                        mass -= self.sum[e, 2 * idx]
                        idx = 2 * idx + 1
                steps[e, i] = idx - self.cap
                weights[e, i] = (self.sum[e, idx] / self.sum[e, 1]) * self.size ** (-beta) / max_weight
        return steps, weights

    def update_priorities(self, idxes, priorities):            # :586-597
        pr = np.asarray(priorities, np.float64).reshape(self.n_envs, self.k)
        for e in range(self.n_envs):
            for idx, p in zip(idxes[e], pr[e]):
                if p == 0:
                    p += 1e-8
                self._set(e, int(idx), p ** self.alpha)
                self.max_priority[e] = max(self.max_priority[e], p)


class MarlBufferOracle:
    """MARL_OffPolicyBuffer (memory_tools_marl.py:634-767) with the agents stacked on one axis: per-agent fields
    [n_envs, n_size, N, ...], state / state_next [n_envs, n_size, S]; bool fields keep their dtype (:717-719, :727-728)."""
    BOOL = ("terminals", "agent_mask", "avail_actions", "avail_actions_next")

    def __init__(self, n_envs, n_size, N, O, A, S):
        self.n_envs, self.n_size = n_envs, n_size
        shapes = dict(obs=(N, O), obs_next=(N, O), actions=(N,), rewards=(N,), terminals=(N,), agent_mask=(N,),
                      state=(S,), state_next=(S,), avail_actions=(N, A), avail_actions_next=(N, A))
        self.data = {k: np.zeros((n_envs, n_size) + v, np.bool_ if k in self.BOOL else np.float32) for k, v in shapes.items()}
        self.ptr = self.size = 0

    def store(self, **step):                                   # :731-740
        for k, v in step.items():
            self.data[k][:, self.ptr] = v
        self.ptr = (self.ptr + 1) % self.n_size
        self.size = min(self.size + 1, self.n_size)

    def sample(self, env_choices, step_choices):               # :742-764 (the two np.random.choice draws, :755-756)
        return {k: v[env_choices, step_choices] for k, v in self.data.items()}


class EpisodeBufferOracle:
    """MARL_OffPolicyBuffer_RNN (memory_tools_marl.py:770-996) with the agents stacked on one axis:
    obs [rows, T+1, N, O], actions/rewards/terminals/agent_mask [rows, T, N], avail_actions [rows, T+1, N, A],
    state [rows, T+1, S], filled [rows, T]; `data` = ring of buffer_size episodes, `episode_data` = one row per env."""

    def __init__(self, n_envs, buffer_size, T, N, O, A, S):
        self.n_envs, self.buffer_size, self.T = n_envs, buffer_size, T
        shapes = dict(obs=(T + 1, N, O), actions=(T, N), rewards=(T, N), terminals=(T, N), agent_mask=(T, N),
                      filled=(T,), state=(T + 1, S), avail_actions=(T + 1, N, A))
        self.shapes = shapes
        self.data = {k: np.zeros((buffer_size,) + v, np.float32) for k, v in shapes.items()}          # clear(), :822-857
        self.ptr = self.size = 0
        self.clear_episodes()

    def clear_episodes(self):                                  # :859-902
        self.episode_data = {k: np.zeros((self.n_envs,) + v, np.float32) for k, v in self.shapes.items()}

    def store(self, episode_steps, **step):                    # :904-921
        e = np.arange(self.n_envs)
        self.episode_data["filled"][e, episode_steps] = 1
        for k, v in step.items():
            self.episode_data[k][e, episode_steps] = v

    def finish_path(self, i_env, episode_step, obs, state, avail_actions):      # :951-968 then store_episodes :923-949
        self.episode_data["state"][i_env, episode_step] = state
        self.episode_data["obs"][i_env, episode_step] = obs
        self.episode_data["avail_actions"][i_env, episode_step] = avail_actions
        for k in self.data:
            self.data[k][self.ptr] = self.episode_data[k][i_env]               # the whole row, stale tail included
        self.ptr = (self.ptr + 1) % self.buffer_size
        self.size = min(self.size + 1, self.buffer_size)
        self.episode_data["filled"][i_env] = 0

    def sample(self, idx):                                     # :970-996 (idx = np.random.choice(size, batch_size))
        return {k: v[idx] for k, v in self.data.items()}


# --------------------------------------------------------------------------------------
# GRU (torch.nn.GRU, one layer, batch_first; third-party arithmetic: PyTorch ATen gru cell --
#   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr); z = sigmoid(W_iz x + b_iz + W_hz h + b_hz);
#   n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (h - n) * z + n
# -- used by Basic_RNN, representations/rnn.py:52-77, built by layers.py:79-98)
# --------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_forward(x, h0, w_ih, w_hh, b_ih, b_hh):
    """x [R,T,I], h0 [R,H] -> hs [R,T,H] + cache for gru_backward."""
    R, T, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype.type
    gi = x @ w_ih.T + b_ih                                               # [R,T,3H]
    hs = np.zeros((R, T, H), x.dtype)
    gates = np.zeros((R, T, 4 * H), x.dtype)                             # r, z, n, (W_hn h + b_hn)
    h = h0.astype(x.dtype)
    for t in range(T):
        gh = h @ w_hh.T + b_hh
        r = _sigmoid(gi[:, t, :H] + gh[:, :H]).astype(x.dtype)
        z = _sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H]).astype(x.dtype)
        n = np.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:]).astype(x.dtype)
        gates[:, t] = np.concatenate([r, z, n, gh[:, 2 * H:]], -1)
        h = ((h - n) * z + n).astype(x.dtype)
        hs[:, t] = h
    return hs, dict(x=x, h0=h0.astype(x.dtype), hs=hs, gates=gates, w_ih=w_ih, w_hh=w_hh)


def gru_backward(cache, dhs):
    """dhs [R,T,H] = d loss / d hs.  Returns (dx [R,T,I], dict of parameter gradients w_ih, w_hh, b_ih, b_hh)."""
    x, hs, gates, w_ih, w_hh = cache["x"], cache["hs"], cache["gates"], cache["w_ih"], cache["w_hh"]
    R, T, _ = x.shape
    H = w_hh.shape[1]
    dgi = np.zeros((R, T, 3 * H), x.dtype)
    dgh = np.zeros((R, T, 3 * H), x.dtype)
    carry = np.zeros((R, H), x.dtype)
    for t in reversed(range(T)):
        r, z, n, hn = (gates[:, t, i * H:(i + 1) * H] for i in range(4))
        hp = hs[:, t - 1] if t > 0 else cache["h0"]
        dh = dhs[:, t] + carry
        dn_pre = dh * (1 - z) * (1 - n * n)
        dz_pre = dh * (hp - n) * z * (1 - z)
        dr_pre = dn_pre * hn * r * (1 - r)
        dgi[:, t] = np.concatenate([dr_pre, dz_pre, dn_pre], -1)
        dgh[:, t] = np.concatenate([dr_pre, dz_pre, dn_pre * r], -1)
        carry = (dh * z + dgh[:, t] @ w_hh).astype(x.dtype)
    hprev = np.concatenate([cache["h0"][:, None], hs[:, :-1]], 1)
    g = dict(w_ih=dgi.reshape(-1, 3 * H).T @ x.reshape(R * T, -1), b_ih=dgi.reshape(-1, 3 * H).sum(0),
             w_hh=dgh.reshape(-1, 3 * H).T @ hprev.reshape(R * T, H), b_hh=dgh.reshape(-1, 3 * H).sum(0))
    return dgi @ w_ih, g


def lstm_forward(x, h0, c0, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM (one layer, batch_first; gate order i | f | g | o): x [R,T,I] -> hs [R,T,H] + cache."""
    R, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.T + b_ih
    hs, cs, gates = np.zeros((R, T, H), x.dtype), np.zeros((R, T + 1, H), x.dtype), np.zeros((R, T, 4 * H), x.dtype)
    h, c = h0.astype(x.dtype), c0.astype(x.dtype)
    cs[:, 0] = c
    for t in range(T):
        pre = gi[:, t] + (h @ w_hh.T + b_hh)
        i, f, o = _sigmoid(pre[:, :H]), _sigmoid(pre[:, H:2 * H]), _sigmoid(pre[:, 3 * H:])
        g = np.tanh(pre[:, 2 * H:3 * H])
        c = (f * c + i * g).astype(x.dtype)
        h = (o * np.tanh(c)).astype(x.dtype)
        gates[:, t] = np.concatenate([i, f, g, o], -1)
        hs[:, t], cs[:, t + 1] = h, c
    return hs, dict(x=x, h0=h0.astype(x.dtype), hs=hs, cs=cs, gates=gates, w_ih=w_ih, w_hh=w_hh, lstm=True)


def lstm_backward(cache, dhs):
    x, hs, cs, gates, w_ih, w_hh = cache["x"], cache["hs"], cache["cs"], cache["gates"], cache["w_ih"], cache["w_hh"]
    R, T, _ = x.shape
    H = w_hh.shape[1]
    dgt = np.zeros((R, T, 4 * H), x.dtype)
    dh_c, dc_c = np.zeros((R, H), x.dtype), np.zeros((R, H), x.dtype)
    for t in reversed(range(T)):
        i, f, g, o = (gates[:, t, q * H:(q + 1) * H] for q in range(4))
        tc = np.tanh(cs[:, t + 1])
        dh = dhs[:, t] + dh_c
        dc = dh * o * (1 - tc * tc) + dc_c
        dgt[:, t] = np.concatenate([dc * g * i * (1 - i), dc * cs[:, t] * f * (1 - f), dc * i * (1 - g * g), dh * tc * o * (1 - o)], -1)
        dh_c = (dgt[:, t] @ w_hh).astype(x.dtype)
        dc_c = (dc * f).astype(x.dtype)
    hprev = np.concatenate([cache["h0"][:, None], hs[:, :-1]], 1)
    flat = dgt.reshape(-1, 4 * H)
    g = dict(w_ih=flat.T @ x.reshape(R * T, -1), b_ih=flat.sum(0), w_hh=flat.T @ hprev.reshape(R * T, H), b_hh=flat.sum(0))
    return dgt @ w_ih, g


def qmix_rnn_agent_forward(sd, prefix, obs, act="relu"):
    """DiscreteActionValueCritic(AgentFeatureEncoder(Basic_RNN)) over whole sequences from zero hidden state
    (base_critics.py:125-132, rnn.py:52-77, iql_learner.py:39-47).  obs [R,T1,O] -> Q [R,T1,A] + caches."""
    rp = f"{prefix}.representation.obs_representation"
    fc_l = collect_seq(sd, f"{rp}.mlp", act, last_act=act)
    q_l = collect_seq(sd, f"{prefix}.critic_head.q_value", act)
    R, T1, O = obs.shape
    fc, q = MLP(fc_l), MLP(q_l)
    f = fc.forward(obs.reshape(R * T1, O)).reshape(R, T1, -1) if fc_l else obs
    w_hh = sd[f"{rp}.rnn.weight_hh_l0"]
    z0 = np.zeros((R, w_hh.shape[1]), obs.dtype)
    if w_hh.shape[0] == 4 * w_hh.shape[1]:                              # rnn: "LSTM" (rnn.py:45-47)
        hs, gc = lstm_forward(f, z0, z0, sd[f"{rp}.rnn.weight_ih_l0"], w_hh, sd[f"{rp}.rnn.bias_ih_l0"], sd[f"{rp}.rnn.bias_hh_l0"])
    else:
        hs, gc = gru_forward(f, z0, sd[f"{rp}.rnn.weight_ih_l0"], w_hh, sd[f"{rp}.rnn.bias_ih_l0"], sd[f"{rp}.rnn.bias_hh_l0"])
    Q = q.forward(hs.reshape(R * T1, -1)).reshape(R, T1, -1)
    return Q, dict(fc=fc, fc_l=fc_l, q=q, q_l=q_l, gru=gc, rp=rp, hs=hs)


def qmix_rnn_forward_backward(sd, batch, cfg, act="relu", group="shared"):
    """QMIX_Learner.update, recurrent branch (qmix_learner.py:24-112 with iql_learner.py:37-83, use_rnn), one group.

    batch (stacked MARL_OffPolicyBuffer_RNN.sample, memory_tools_marl.py:970-996): obs [B,N,T+1,O], actions [B,N,T],
    rewards [B,N,T], terminals [B,N,T], agent_mask [B,N,T], avail_actions [B,N,T+1,A] (when use_actions_mask),
    state [B,T+1,S], filled [B,T].
    cfg["agent_grad"] (default False = the unmodified reference): iql_learner.py:58 re-slices q_eval inside
    torch.no_grad(), so the agent networks get NO gradient (only the mixer trains); True = gradient flows through the
    Q head, the GRU (BPTT) and the fc layer.  The action mask of step t+1 is applied on the TIME axis (the reference's
    iql_learner.py:78 slices the agent axis and raises IndexError).  Both deviations are pinned by the `_fixed`
    fixture, see oracle/make_golden.py golden_qmix_rnn."""
    dt = np.float32
    pe, pt = f"individual_q_networks.{group}", f"target_individual_q_networks.{group}"
    B, N, T1, O = batch["obs"].shape
    T = T1 - 1
    obs = batch["obs"].reshape(B * N, T1, O).astype(dt)
    Q, c = qmix_rnn_agent_forward(sd, pe, obs, act)                      # iql_learner.py:41-47
    A = Q.shape[-1]
    Qt, _ = qmix_rnn_agent_forward(sd, pt, obs, act)                     # :53-57
    use_mask = cfg.get("use_actions_mask", True)
    avail = batch["avail_actions"].reshape(B * N, T1, A) if use_mask else None
    qd = Q.copy()
    if use_mask:
        qd[avail == 0] = -1e10                                           # value_factorization.py:87-90
    a_next = qd.argmax(-1)[:, 1:]                                        # iql_learner.py:51,60
    q_eval, q_next = Q[:, :-1], Qt[:, 1:].copy()                         # :58-59
    if use_mask:
        q_next[avail[:, 1:] == 0] = -1e10                                # :76-81 (time axis)
    rewards_tot = batch["rewards"].astype(dt).mean(1)                    # qmix_learner.py:34  [B,T]
    terminals_tot = batch["terminals"].astype(bool).all(1).astype(dt)    # :35
    filled = batch["filled"].astype(dt)                                  # [B,T]
    mask = (batch["agent_mask"].astype(dt) * filled[:, None, :]).reshape(B * N, T)   # outputs.py:138-143
    a_taken = batch["actions"].reshape(B * N, T).astype(np.int64)
    q_eval_taken = np.take_along_axis(q_eval, a_taken[..., None], -1)[..., 0]        # :48-50
    if cfg.get("double_q", True):
        q_next_taken = np.take_along_axis(q_next, a_next[..., None], -1)[..., 0]     # :52-55
    else:
        q_next_taken = q_next.max(-1)                                    # :57-58
    qe = (q_eval_taken * mask).reshape(B, N, T).transpose(0, 2, 1).reshape(B * T, N)  # :60,64-66 + Q_tot reshape
    qn = (q_next_taken * mask).reshape(B, N, T).transpose(0, 2, 1).reshape(B * T, N)
    S = batch["state"].shape[-1]
    q_tot_eval, cache = qmix_mixer_forward(sd, "eval_Qtot", qe, batch["state"][:, :-1].reshape(B * T, S).astype(dt))   # :69
    q_tot_next, _ = qmix_mixer_forward(sd, "target_Qtot", qn, batch["state"][:, 1:].reshape(B * T, S).astype(dt))      # :70
    target = rewards_tot.reshape(-1) + (1 - terminals_tot.reshape(-1)) * dt(cfg["gamma"]) * q_tot_next                 # :78
    fl = filled.reshape(-1)
    td = (q_tot_eval - target) * fl                                      # :83
    loss = (td ** 2).sum() / fl.sum()                                    # :84
    dq_tot = (2 * td * fl / fl.sum()).astype(dt)
    dq_m, grads = qmix_mixer_backward(sd, "eval_Qtot", cache, dq_tot)
    info = dict(q_eval=Q, q_target=Qt, q_tot_eval=q_tot_eval, q_tot_next=q_tot_next, q_tot_target=target, loss=loss,
                predictQ=q_tot_eval.mean(), hs=c["hs"], actions_next=a_next)
    if not cfg.get("agent_grad", False):
        return info, grads
    dq_taken = dq_m.reshape(B, T, N).transpose(0, 2, 1).reshape(B * N, T) * mask
    dQ = np.zeros((B * N, T1, A), dt)
    np.put_along_axis(dQ[:, :-1], a_taken[..., None], dq_taken[..., None], -1)
    dhs, g_q = c["q"].backward(dQ.reshape(B * N * T1, A), need_dx=True)
    for L, (gw, gb) in zip(c["q_l"], g_q):
        grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    df, gg = (lstm_backward if c["gru"].get("lstm") else gru_backward)(c["gru"], dhs.reshape(B * N, T1, -1))
    rp = c["rp"]
    grads[f"{rp}.rnn.weight_ih_l0"], grads[f"{rp}.rnn.weight_hh_l0"] = gg["w_ih"], gg["w_hh"]
    grads[f"{rp}.rnn.bias_ih_l0"], grads[f"{rp}.rnn.bias_hh_l0"] = gg["b_ih"], gg["b_hh"]
    if c["fc_l"]:
        _, g_fc = c["fc"].backward(df.reshape(B * N * T1, -1), need_dx=False)
        for L, (gw, gb) in zip(c["fc_l"], g_fc):
            grads[L["name"] + ".weight"], grads[L["name"] + ".bias"] = gw, gb
    return info, grads


def qmix_copy_target(sd):                                               # value_factorization.py:169-174
    for k in list(sd):
        if k.startswith("individual_q_networks."):
            sd["target_" + k][...] = sd[k]
        elif k.startswith("eval_Qtot."):
            sd["target_Qtot." + k[len("eval_Qtot."):]][...] = sd[k]


# --------------------------------------------------------------------------------------
# Sampling helpers (inverse-CDF with supplied uniforms; parity is defined on fixed uniforms)
# --------------------------------------------------------------------------------------
def categorical_sample_icdf(logits, u):
    """Smallest k with cumsum(softmax(logits))[k] > u  (float32 cumulative sum, left to right)."""
    p = np.exp(log_softmax(logits.astype(np.float32))).astype(np.float32)
    c = np.cumsum(p, axis=-1, dtype=np.float32)
    k = (c <= u[:, None]).sum(-1)
    return np.minimum(k, logits.shape[-1] - 1)


def gaussian_sample_reparam(mu, log_std, z):
    """Normal(mu, exp(log_std)).sample() with supplied standard normals z and its summed log-prob
    (DiagGaussianDistribution.stochastic_sample / log_prob, distributions.py:172-180): x = mu + std * z."""
    mu, z = mu.astype(np.float32), z.astype(np.float32)
    ls = log_std.astype(np.float32)
    std = np.exp(ls)
    x = (mu + std * z).astype(np.float32)
    lp = (-((x - mu) ** 2) / (2 * std * std) - ls - np.float32(math.log(math.sqrt(2 * math.pi)))).sum(-1)
    return x, lp.astype(np.float32)


# --------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al., SC'11) as the device uses it (csrc/rng.h): key = 64-bit seed, counter (c0, c1, c2, 0).
# OUR engine's random streams, not the reference's (torch / NumPy generators cannot be matched): restated so that a
# whole device rollout -- reset states and sampled actions included -- can be replayed on the CPU.
# --------------------------------------------------------------------------------------
STREAM_ACTION, STREAM_GAUSS, STREAM_RESET_A, STREAM_RESET_B = 0x41435431, 0x47415500, 0x52455345, 0x52455346


def philox4x32(seed, c0, c1, c2, c3=0):
    """Vectorised over the counter words (uint32 arrays or scalars; the device always uses c3 = 0); returns four uint32
    arrays.  Pinned by the three known-answer vectors of Random123's kat_vectors (tests/test_cpu_host.py)."""
    c0, c1, c2, c3 = np.broadcast_arrays(np.asarray(c0, np.uint64), np.asarray(c1, np.uint64), np.asarray(c2, np.uint64),
                                         np.asarray(c3, np.uint64))
    c = [c0 & 0xFFFFFFFF, c1 & 0xFFFFFFFF, c2 & 0xFFFFFFFF, c3 & 0xFFFFFFFF]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return [x.astype(np.uint32) for x in c]


def u01(x):
    """rng.h u01: top 24 bits -> [0, 1) float32."""
    return ((np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def u01d(a, b):
    a, b = np.asarray(a, np.uint64), np.asarray(b, np.uint64)
    return (((a << np.uint64(21)) ^ b) & np.uint64((1 << 53) - 1)).astype(np.float64) * (1.0 / 9007199254740992.0)


def cartpole_reset_state(seed, envs, episodes):
    """cartpole.h cartpole_reset: uniform(-0.05, 0.05) float64 state of env `e` at the start of its `episode`-th episode."""
    r = philox4x32(seed, envs, episodes, STREAM_RESET_A)
    q = philox4x32(seed, envs, episodes, STREAM_RESET_B)
    return np.stack([-0.05 + 0.1 * u01d(r[j], q[j]) for j in range(4)], -1)


def classic_reset_state(kind, seed, envs, episodes):
    """classic.h classic_reset: float64 initial state [n, 4] of env `e` at the start of its `episode`-th episode.  kind 1 Pendulum-v1
    (uniform(-[pi, 1], [pi, 1])), 2 MountainCar-v0 (position uniform(-0.6, -0.4), velocity 0), 3 Acrobot-v1 (uniform(-0.1, 0.1) x 4):
    the ranges of Gymnasium's reset(); the random words are the engine's Philox streams."""
    r = philox4x32(seed, envs, episodes, STREAM_RESET_A)
    q = philox4x32(seed, envs, episodes, STREAM_RESET_B)
    u = [u01d(r[j], q[j]) for j in range(4)]
    z = np.zeros_like(u[0])
    if kind == 1:
        return np.stack([-math.pi + 2.0 * math.pi * u[0], -1.0 + 2.0 * u[1], z, z], -1)
    if kind == 2:
        return np.stack([-0.6 + 0.2 * u[0], z, z, z], -1)
    return np.stack([-0.1 + 0.2 * u[j] for j in range(4)], -1)


def action_uniforms(seed, n_envs, step):
    """The uniform the device draws for env e at global vector step `step` (xrl_policy_sample / the fused rollout kernels)."""
    return u01(philox4x32(seed, np.arange(n_envs), step, STREAM_ACTION)[0])


def action_gaussians(seed, n_envs, step, A):
    """The standard normals the device draws for env e, action dim j at global vector step `step` (xrl_policy_sample,
    xrl_wide_act_step: Box-Muller on the first two words of Philox(seed; e, step, STREAM_GAUSS + j), float32)."""
    z = np.zeros((n_envs, A), np.float32)
    for j in range(A):
        r = philox4x32(seed, np.arange(n_envs), step, STREAM_GAUSS + j)
        u1, u2 = np.maximum(u01(r[0]), np.float32(5.96e-8)), u01(r[1])
        z[:, j] = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return z


# --------------------------------------------------------------------------------------
# Pendulum-v1, MountainCar-v0, Acrobot-v1 (Gymnasium classic_control: pendulum.py, mountain_car.py, acrobot.py; the reference
# pins gymnasium >= 0.28, < 1.3 in setup.py:73; third-party, not in the reference tree and not in this image).  PARITY UNPINNED: no
# Gymnasium here to check these restatements of the published equations against -- they pin the DEVICE envs (csrc/classic.h) to
# one NumPy statement of the same equations.  float64 state, float32 observations, vectorised over envs.
# --------------------------------------------------------------------------------------
class PendulumOracle:
    max_speed, max_torque, dt, g, m, l, max_steps = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0, 200

    def __init__(self, state):
        self.state = np.array(state, np.float64)[:, :2].copy()      # theta, theta_dot
        self.steps = np.zeros(len(self.state), np.int64)

    @staticmethod
    def observe(state):
        return np.stack([np.cos(state[:, 0]), np.sin(state[:, 0]), state[:, 1]], 1).astype(np.float32)

    def step(self, action):
        th, thdot = self.state.T
        u = np.clip(np.asarray(action, np.float32).reshape(-1), np.float32(-self.max_torque), np.float32(self.max_torque)).astype(np.float64)
        an = ((th + math.pi) % (2 * math.pi)) - math.pi                  # angle_normalize
        costs = an ** 2 + 0.1 * thdot ** 2 + 0.001 * (u ** 2)
        nthdot = thdot + (3 * self.g / (2 * self.l) * np.sin(th) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
        nthdot = np.clip(nthdot, -self.max_speed, self.max_speed)
        nth = th + nthdot * self.dt
        self.state = np.stack([nth, nthdot], 1)
        self.steps += 1
        return self.observe(self.state), (-costs).astype(np.float32), np.zeros(len(th), bool), self.steps >= self.max_steps


class MountainCarOracle:
    min_position, max_position, max_speed, goal_position, force, gravity, max_steps = -1.2, 0.6, 0.07, 0.5, 0.001, 0.0025, 200

    def __init__(self, state):
        self.state = np.array(state, np.float64)[:, :2].copy()      # position, velocity
        self.steps = np.zeros(len(self.state), np.int64)

    @staticmethod
    def observe(state):
        return state.astype(np.float32)

    def step(self, action):
        position, velocity = self.state.T.copy()
        velocity = velocity + (np.asarray(action, np.float64) - 1) * self.force + np.cos(3 * position) * (-self.gravity)
        velocity = np.clip(velocity, -self.max_speed, self.max_speed)
        position = position + velocity
        position = np.clip(position, self.min_position, self.max_position)
        velocity = np.where((position == self.min_position) & (velocity < 0), 0.0, velocity)
        self.state = np.stack([position, velocity], 1)
        self.steps += 1
        term = (position >= self.goal_position) & (velocity >= 0.0)
        return self.observe(self.state), np.full(len(position), -1.0, np.float32), term, self.steps >= self.max_steps


class AcrobotOracle:
    dt, max_steps = 0.2, 500

    def __init__(self, state):
        self.state = np.array(state, np.float64).copy()             # theta1, theta2, dtheta1, dtheta2
        self.steps = np.zeros(len(self.state), np.int64)

    @staticmethod
    def observe(s):
        return np.stack([np.cos(s[:, 0]), np.sin(s[:, 0]), np.cos(s[:, 1]), np.sin(s[:, 1]), s[:, 2], s[:, 3]], 1).astype(np.float32)

    @staticmethod
    def _dsdt(s, a):                                                # acrobot.py: _dsdt, "book" variant
        m1 = m2 = l1 = 1.0
        lc1 = lc2 = 0.5
        I1 = I2 = 1.0
        g = 9.8
        t1, t2, dt1, dt2 = s.T
        d1 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * np.cos(t2)) + I1 + I2
        d2 = m2 * (lc2 ** 2 + l1 * lc2 * np.cos(t2)) + I2
        phi2 = m2 * lc2 * g * np.cos(t1 + t2 - math.pi / 2.0)
        phi1 = (-m2 * l1 * lc2 * dt2 ** 2 * np.sin(t2) - 2 * m2 * l1 * lc2 * dt2 * dt1 * np.sin(t2)
                + (m1 * lc1 + m2 * l1) * g * np.cos(t1 - math.pi / 2) + phi2)
        ddt2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * dt1 ** 2 * np.sin(t2) - phi2) / (m2 * lc2 ** 2 + I2 - d2 ** 2 / d1)
        ddt1 = -(d2 * ddt2 + phi1) / d1
        return np.stack([dt1, dt2, ddt1, ddt2], 1)

    @staticmethod
    def _wrap(x, m, M):
        x = x.copy()
        diff = M - m
        while (x > M).any():
            x = np.where(x > M, x - diff, x)
        while (x < m).any():
            x = np.where(x < m, x + diff, x)
        return x

    def step(self, action):
        s, dt = self.state, self.dt
        a = np.asarray(action, np.float64) - 1.0                     # AVAIL_TORQUE = [-1, 0, +1]
        k1 = self._dsdt(s, a)
        k2 = self._dsdt(s + dt / 2.0 * k1, a)
        k3 = self._dsdt(s + dt / 2.0 * k2, a)
        k4 = self._dsdt(s + dt * k3, a)
        ns = s + dt / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        ns[:, 0] = self._wrap(ns[:, 0], -math.pi, math.pi)
        ns[:, 1] = self._wrap(ns[:, 1], -math.pi, math.pi)
        ns[:, 2] = np.clip(ns[:, 2], -4 * math.pi, 4 * math.pi)
        ns[:, 3] = np.clip(ns[:, 3], -9 * math.pi, 9 * math.pi)
        self.state = ns
        self.steps += 1
        term = -np.cos(ns[:, 0]) - np.cos(ns[:, 1] + ns[:, 0]) > 1.0
        return self.observe(ns), np.where(term, 0.0, -1.0).astype(np.float32), term, self.steps >= self.max_steps


# --------------------------------------------------------------------------------------
# CartPole-v1 physics (public equations, Barto-Sutton-Anderson 1983 / Gymnasium classic_control;
# NOT part of the reference tree -- SURVEY.md section 7).  float64 state, float32 observations.
# --------------------------------------------------------------------------------------
class CartPoleOracle:
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    theta_thr, x_thr, max_steps = 12 * 2 * math.pi / 360, 2.4, 500

    def __init__(self, state):
        self.state = np.array(state, np.float64)      # [n,4]
        self.steps = np.zeros(len(self.state), np.int64)

    def step(self, action):
        x, xd, th, thd = self.state.T
        force = np.where(np.asarray(action) == 1, self.force_mag, -self.force_mag)
        ct, st = np.cos(th), np.sin(th)
        total_mass = self.masspole + self.masscart
        pml = self.masspole * self.length
        temp = (force + pml * thd * thd * st) / total_mass
        thacc = (self.gravity * st - ct * temp) / (self.length * (4.0 / 3.0 - self.masspole * ct * ct / total_mass))
        xacc = temp - pml * thacc * ct / total_mass
        x = x + self.tau * xd
        xd = xd + self.tau * xacc
        th = th + self.tau * thd
        thd = thd + self.tau * thacc
        self.state = np.stack([x, xd, th, thd], 1)
        self.steps += 1
        term = (x < -self.x_thr) | (x > self.x_thr) | (th < -self.theta_thr) | (th > self.theta_thr)
        trunc = self.steps >= self.max_steps
        return self.state.astype(np.float32), np.ones(len(x), np.float32), term, trunc
