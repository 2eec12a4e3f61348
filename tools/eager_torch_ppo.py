"""MEASUREMENT INFRASTRUCTURE -- the "un-fused GPU" comparator of BASELINE.md section 2: the same PPO-Clip CartPole loop
(256 envs x 256 steps, obs / reward normalisation, GAE(0.98, 0.95), 8 epochs x 8 minibatches, clip 0.5, Adam eps 1e-5)
written with STOCK PyTorch-ROCm ops on the same MI355X: nn.Linear + autograd + torch.distributions.Categorical +
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, a CartPole vectorised with torch tensor ops, everything resident on
the device (no per-step host transfers: the friendliest un-fused arrangement; the reference itself moves every step's
arrays through NumPy).  It is what an engineer gets from "put the reference's math on the GPU with eager PyTorch";
bench.py reports its env-steps/s beside the HIP engine's.  Not a parity artefact (its RNG streams and the order of its
running-statistics merges differ); never imported by xuance_amd/."""
import math
import time

import torch
from torch import nn


class TorchCartPole:
    """CartPole-v1 (Barto-Sutton-Anderson, Gymnasium constants) for n envs as float64 tensor ops, auto-reset."""

    def __init__(self, n, device, seed=1, max_steps=500):
        self.n, self.dev, self.max_steps = n, device, max_steps
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.state = self._fresh(n)
        self.steps = torch.zeros(n, dtype=torch.int32, device=device)

    def _fresh(self, k):
        return torch.rand(k, 4, dtype=torch.float64, device=self.dev, generator=self.gen) * 0.1 - 0.05

    def obs(self):
        return self.state.float()

    def step(self, action):
        x, xd, th, thd = self.state.unbind(1)
        force = torch.where(action == 1, 10.0, -10.0).double()
        ct, st = torch.cos(th), torch.sin(th)
        temp = (force + 0.05 * thd * thd * st) / 1.1
        thacc = (9.8 * st - ct * temp) / (0.5 * (4.0 / 3.0 - 0.1 * ct * ct / 1.1))
        xacc = temp - 0.05 * thacc * ct / 1.1
        x, xd, th, thd = x + 0.02 * xd, xd + 0.02 * xacc, th + 0.02 * thd, thd + 0.02 * thacc
        nxt = torch.stack([x, xd, th, thd], 1)
        self.steps += 1
        term = (x.abs() > 2.4) | (th.abs() > 12 * 2 * math.pi / 360)
        trunc = (self.steps >= self.max_steps) & ~term
        done = term | trunc
        self.state = torch.where(done[:, None], self._fresh(self.n), nxt)
        self.steps = torch.where(done, torch.zeros_like(self.steps), self.steps)
        return nxt.float(), term, trunc, done


class RunningMeanStd:
    def __init__(self, shape, device):
        self.mean = torch.zeros(shape, device=device)
        self.var = torch.ones(shape, device=device)
        self.count = torch.full((), 1e-4, dtype=torch.float64, device=device)

    def update_moments(self, bm, bv, bc):                    # statistic_tools.py:149-185
        delta = bm - self.mean
        tot = self.count + bc
        self.mean = self.mean + delta * (bc / tot).float()
        m2 = self.var * self.count.float() + bv * bc.float() + delta * delta * (self.count * bc / tot).float()
        self.var = m2 / tot.float()
        self.count = tot


def mlp(sizes, act, last_act=None):
    layers = []
    for i in range(len(sizes) - 1):
        lin = nn.Linear(sizes[i], sizes[i + 1])
        nn.init.orthogonal_(lin.weight)
        nn.init.zeros_(lin.bias)
        layers.append(lin)
        if i < len(sizes) - 2:
            layers.append(act())
    return nn.Sequential(*layers)


class EagerPPO:
    def __init__(self, n_envs=256, horizon=256, device="cuda", seed=1):
        torch.manual_seed(seed)
        self.n, self.T, self.dev = n_envs, horizon, device
        self.rep = nn.Sequential(nn.Linear(4, 128), nn.LeakyReLU()).to(device)
        nn.init.orthogonal_(self.rep[0].weight); nn.init.zeros_(self.rep[0].bias)
        self.actor = mlp([128, 128, 2], nn.LeakyReLU).to(device)
        self.critic = mlp([128, 128, 1], nn.LeakyReLU).to(device)
        self.params = list(self.rep.parameters()) + list(self.actor.parameters()) + list(self.critic.parameters())
        self.opt = torch.optim.Adam(self.params, 4e-4, eps=1e-5)
        self.env = TorchCartPole(n_envs, device, seed)
        self.obs_rms, self.ret_rms = RunningMeanStd((4,), device), RunningMeanStd((), device)
        self.ret_track = torch.zeros(n_envs, device=device)
        T, n = horizon, n_envs
        z = lambda *s: torch.zeros(*s, device=device)
        self.buf = dict(obs=z(T, n, 4), act=z(T, n), rew=z(T, n), val=z(T, n), logp=z(T, n), term=z(T, n), end=z(T, n), boot=z(T, n))
        self.gamma, self.lam = 0.98, 0.95

    def _norm(self, x):
        return ((x - self.obs_rms.mean) / (self.obs_rms.var.sqrt() + 1e-8)).clamp(-5, 5)

    @torch.no_grad()
    def rollout(self):
        b, n = self.buf, self.n
        obs = self.env.obs()
        for t in range(self.T):
            self.obs_rms.update_moments(obs.mean(0), obs.var(0, unbiased=False), torch.tensor(float(n), dtype=torch.float64, device=self.dev))
            x = self._norm(obs)
            h = self.rep(x)
            dist = torch.distributions.Categorical(logits=self.actor(h))
            a = dist.sample()
            nxt, term, trunc, done = self.env.step(a)
            rstd = self.ret_rms.var.sqrt().clamp(0.1, 100)
            b["obs"][t], b["act"][t], b["val"][t], b["logp"][t] = x, a.float(), self.critic(h)[:, 0], dist.log_prob(a)
            b["rew"][t] = (1.0 / rstd).clamp(-5, 5).expand(n)
            b["term"][t] = term.float()
            b["end"][t] = (done | torch.tensor(t == self.T - 1, device=self.dev)).float()
            b["boot"][t] = self.critic(self.rep(self._norm(nxt)))[:, 0]
            self.ret_track = self.gamma * self.ret_track + 1.0
            # merge the returns of the episodes that ended in this step (batch moments; no host synchronisation)
            d = done.float()
            k = d.sum()
            fm = (self.ret_track * d).sum() / k.clamp(min=1.0)
            fv = (((self.ret_track - fm) ** 2) * d).sum() / k.clamp(min=1.0)
            self.ret_rms.update_moments(fm, fv, k.double())
            self.ret_track = torch.where(done, torch.zeros_like(self.ret_track), self.ret_track)
            obs = self.env.obs()
        # GAE, vectorised over envs, sequential in t (memory_tools.py:242-265 with the per-path bootstrap values)
        adv = torch.zeros_like(b["rew"])
        last = torch.zeros(n, device=self.dev)
        for t in reversed(range(self.T)):
            end, term = b["end"][t], b["term"][t]
            nv = b["val"][t + 1] if t + 1 < self.T else b["boot"][t]
            nv = torch.where(end > 0, b["boot"][t], nv)
            last = torch.where(end > 0, torch.zeros_like(last), last)
            delta = b["rew"][t] + (1 - term) * self.gamma * nv - b["val"][t]
            last = delta + (1 - term) * self.gamma * self.lam * last
            adv[t] = last
        b["adv"], b["ret"] = adv, adv + b["val"]

    def update(self, n_epochs=8, n_minibatch=8):
        b, N = self.buf, self.n * self.T
        flat = {k: b[k].reshape(N, -1).squeeze(-1) for k in ("obs", "act", "ret", "adv", "logp")}
        flat["obs"] = b["obs"].reshape(N, 4)
        bs = N // n_minibatch
        info = {}
        for _ in range(n_epochs):
            perm = torch.randperm(N, device=self.dev)
            for k in range(n_minibatch):
                idx = perm[k * bs:(k + 1) * bs]
                obs, act, ret, adv, old = (flat[q][idx] for q in ("obs", "act", "ret", "adv", "logp"))
                adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
                h = self.rep(obs)
                dist = torch.distributions.Categorical(logits=self.actor(h))
                v = self.critic(h)[:, 0]
                ratio = (dist.log_prob(act) - old).exp()
                a_loss = -torch.minimum(ratio.clamp(0.8, 1.2) * adv, ratio * adv).mean()
                c_loss = ((v - ret) ** 2).mean()
                loss = a_loss - 0.01 * dist.entropy().mean() + 0.25 * c_loss
                self.opt.zero_grad()
                loss.backward()
                nn.utils.clip_grad_norm_(self.params, 0.5)
                self.opt.step()
        info["loss"] = float(loss.detach())                              # the one host read per update phase
        return info


def measure(n_envs=256, horizon=256, steps=3, warmup=1, device="cuda"):
    agent = EagerPPO(n_envs, horizon, device)
    for _ in range(warmup):
        agent.rollout(); agent.update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.rollout()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        agent.update()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(n_envs * horizon * steps / dt, 1), "unit": "env-steps/s", "ms_per_step": round(dt / steps * 1e3, 2),
            "kind": "stock PyTorch-ROCm eager ops (nn.Linear + autograd + torch.optim.Adam), same loop, same MI355X, device-resident",
            "sample": "%d rollout+update steps of %d envs x %d after %d warm-up (tools/eager_torch_ppo.py)" % (steps, n_envs, horizon, warmup)}


if __name__ == "__main__":
    import json
    print(json.dumps(measure()))
