"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, agi-brain/xuance v1.4.4) on CPU through oracle/ref_shim.py.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
The reference does not exist on the GPU box, so the resulting fixtures are committed.
Every fixture stores the exact inputs next to the reference's outputs so any engine can replay them.

Versions used to generate the committed fixtures: torch 2.10.0+rocm7.0 (CPU path), numpy 2.2.6,
python 3.10.12, 8 threads.
"""
import copy
import os
import sys
from argparse import Namespace

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
from torch import nn  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

from xuance.common.memory_tools import DummyOnPolicyBuffer, DummyOffPolicyBuffer, DummyOffPolicyBuffer_Atari  # noqa: E402
from xuance.common.statistic_tools import RunningMeanStd  # noqa: E402
from xuance.common.callback import BaseCallback  # noqa: E402
from xuance.common import AgentGrouping  # noqa: E402
from xuance.torch.learners import PPO_Learner, A2C_Learner, DQN_Learner, DDQN_Learner, QMIX_Learner  # noqa: E402
from xuance.torch.rl_models.representations import Basic_MLP, Basic_Identical, Basic_CNN  # noqa: E402
from xuance.torch.rl_models.heads.actor_head import CategoricalActorHead, GaussianActorHead  # noqa: E402
from xuance.torch.rl_models.heads.critic_head import ValueHead  # noqa: E402
from xuance.torch.rl_models.heads.q_mix_head import QMIX_Mixer  # noqa: E402
from xuance.torch.rl_models.architectures.single_agent.actor_critic import SharedActorCritic  # noqa: E402
from xuance.torch.rl_models.architectures.single_agent.deep_q_network import DeepQNetwork  # noqa: E402
from xuance.torch.rl_models.architectures.multi_agent.value_factorization import MixingQNetwork  # noqa: E402

sp = ref_shim.spaces()


class Capture(BaseCallback):
    """Records every tensor kwarg handed to on_update_end (ppo_learner.py:90-94 etc.)."""

    def __init__(self):
        super().__init__()
        self.records = []

    def on_update_end(self, iterations, **kwargs):
        rec = {}
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                rec[k] = v.detach().cpu().numpy().copy()
        self.records.append(rec)
        return {}


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def flat(prefix, d):
    return {f"{prefix}/{k}": np.asarray(v) for k, v in d.items()}


def base_config(**kw):
    c = dict(distributed_training=False, episode_length=60, learning_rate=4e-4, gamma=0.98, use_grad_clip=True,
             grad_clip_norm=0.5, device="cpu", model_dir="/tmp/xrl_golden_models", running_steps=120000,
             parallels=4)
    c.update(kw)
    return Namespace(**c)


# ------------------------------------------------------------------------------ on-policy buffer
def golden_onpolicy_buffer():
    rng = np.random.default_rng(7)
    n_envs, T, D = 6, 24, 4
    out = {}
    for tag, use_gae in (("gae", True), ("nogae", False)):
        buf = DummyOnPolicyBuffer(sp.Box(-1, 1, (D,)), sp.Discrete(2), {"old_logp": ()}, n_envs, T,
                                  use_gae=use_gae, use_advnorm=True, gamma=0.98, gae_lam=0.95)
        obs = rng.standard_normal((T, n_envs, D)).astype(np.float32)
        act = rng.integers(0, 2, (T, n_envs)).astype(np.float32)
        rew = rng.standard_normal((T, n_envs)).astype(np.float32)
        val = rng.standard_normal((T, n_envs)).astype(np.float32)
        logp = (-rng.random((T, n_envs))).astype(np.float32)
        term = rng.random((T, n_envs)) < 0.08
        trunc = (rng.random((T, n_envs)) < 0.06) & ~term
        boot = rng.standard_normal((T, n_envs)).astype(np.float32)   # V(next_obs) the agent would pass
        idx = rng.permutation(n_envs * T)[:48]
        for t in range(T):
            buf.store(obs[t], act[t], rew[t], val[t], term[t], {"old_logp": logp[t]})
            if buf.full:   # ppo_agent.py:129-142: finish every env, train (sample), then clear()
                for i in range(n_envs):
                    buf.finish_path(0.0 if term[t, i] else boot[t, i], i)
                returns, advantages = buf.returns.copy(), buf.advantages.copy()
                s = buf.sample(idx)
                buf.clear()
            for i in range(n_envs):  # ppo_agent.py:146-157 (after a clear() these act on an empty slice)
                if term[t, i] or trunc[t, i]:
                    buf.finish_path(0.0 if term[t, i] else boot[t, i], i)
        assert buf.ptr == 0 and buf.size == 0 and not buf.returns.any()
        out.update(flat(tag, dict(obs=obs, act=act, rew=rew, val=val, logp=logp, term=term, trunc=trunc, boot=boot,
                                  returns=returns, advantages=advantages, idx=idx,
                                  s_obs=s["obs"], s_actions=s["actions"], s_returns=s["returns"],
                                  s_values=s["values"], s_old_logp=s["aux_batch"]["old_logp"],
                                  s_advantages=s["advantages"])))
    out["meta"] = np.array([n_envs, T, D, 0.98, 0.95])
    np.savez_compressed(os.path.join(OUT, "onpolicy_buffer.npz"), **out)


# ------------------------------------------------------------------------------ off-policy buffer
def golden_offpolicy_buffer():
    rng = np.random.default_rng(11)
    out = {}
    for tag, cls, shape, dtype in (("f32", DummyOffPolicyBuffer, (5,), np.float32),
                                   ("u8", DummyOffPolicyBuffer_Atari, (12, 12, 4), np.uint8)):
        n_envs, n_size, bs, steps = 4, 10, 16, 13   # wraps around the ring
        buf = cls(sp.Box(0, 255, shape), sp.Discrete(4), None, n_envs, n_envs * n_size, bs)
        if dtype == np.uint8:
            obs = rng.integers(0, 256, (steps, n_envs) + shape).astype(np.uint8)
            nxt = rng.integers(0, 256, (steps, n_envs) + shape).astype(np.uint8)
        else:
            obs = rng.standard_normal((steps, n_envs) + shape).astype(np.float32)
            nxt = rng.standard_normal((steps, n_envs) + shape).astype(np.float32)
        act = rng.integers(0, 4, (steps, n_envs))
        rew = rng.standard_normal((steps, n_envs)).astype(np.float32)
        term = rng.random((steps, n_envs)) < 0.1
        for t in range(steps):
            buf.store(obs[t], act[t], rew[t], term[t], nxt[t])
        np.random.seed(123)
        env = np.random.choice(n_envs, bs)
        step = np.random.choice(buf.size, bs)
        np.random.seed(123)
        s = buf.sample()
        out.update(flat(tag, dict(obs=obs, nxt=nxt, act=act, rew=rew, term=term, env=env, step=step,
                                  s_obs=s["obs"], s_actions=s["actions"], s_obs_next=s["obs_next"],
                                  s_rewards=s["rewards"], s_terminals=s["terminals"],
                                  meta=np.array([n_envs, n_size, bs, steps, buf.ptr, buf.size]))))
    np.savez_compressed(os.path.join(OUT, "offpolicy_buffer.npz"), **out)


# ------------------------------------------------------------------------------ running mean/std
def golden_rms():
    rng = np.random.default_rng(3)
    n, D, steps = 16, 4, 12
    rms = RunningMeanStd((D,))
    xs = (rng.standard_normal((steps, n, D)) * np.array([1, 3, 0.1, 10]) + np.array([0, 1, -2, 5])).astype(np.float32)
    means, vars_, counts, normed = [], [], [], []
    for t in range(steps):
        rms.update(xs[t])                                              # ppo_agent.py:114
        means.append(rms.mean.copy()); vars_.append(rms.var.copy()); counts.append(rms.count)
        normed.append(np.clip((xs[t] - rms.mean) / (rms.std + 1e-8), -5, 5))   # agent.py:262-283
    # return-normaliser: sequential single-sample updates (ppo_agent.py:144-149)
    ret = RunningMeanStd(())
    rs = rng.standard_normal(20).astype(np.float32) * 30
    rmean, rvar, rcount, rproc = [], [], [], []
    rew = rng.standard_normal((20, 3)).astype(np.float32) * 4
    for i in range(20):
        ret.update(rs[i:i + 1])
        rmean.append(ret.mean.copy()); rvar.append(ret.var.copy()); rcount.append(ret.count)
        std = np.clip(ret.std, 0.1, 100)
        rproc.append(np.clip(rew[i] / std, -5, 5))
    np.savez_compressed(os.path.join(OUT, "rms.npz"), xs=xs, means=np.array(means), vars=np.array(vars_),
                        counts=np.array(counts), normed=np.array(normed), rs=rs, rmean=np.array(rmean),
                        rvar=np.array(rvar), rcount=np.array(rcount), rew=rew, rproc=np.array(rproc))


# ------------------------------------------------------------------------------ PPO
def run_learner_updates(learner, model, cb, batches, call):
    out = {}
    out.update(flat("init", sd_np(model)))
    for u, b in enumerate(batches):
        info = call(b)
        out.update(flat(f"u{u}/batch", {k: v for k, v in b.items()}))
        out.update(flat(f"u{u}/info", {k: np.float64(v) for k, v in info.items() if np.isscalar(v) or
                                       isinstance(v, (float, int, torch.Tensor))}))
        out.update(flat(f"u{u}/cb", cb.records[-1]))
        out.update(flat(f"u{u}/grad", {n: p.grad.detach().numpy().copy()
                                       for n, p in model.named_parameters() if p.grad is not None}))
        out.update(flat(f"u{u}/param", sd_np(model)))
    opt = learner.optimizer
    if isinstance(opt, dict):                                    # IQL: one optimiser per group (iql_learner.py:24-36)
        opt = next(iter(opt.values()))
    names = [n for n, _ in model.named_parameters()]
    for n, p in model.named_parameters():
        st = opt.state.get(p, None)
        if st:
            out[f"adam/exp_avg/{n}"] = st["exp_avg"].numpy().copy()
            out[f"adam/exp_avg_sq/{n}"] = st["exp_avg_sq"].numpy().copy()
    out["param_names"] = np.array(names)
    return out


def q16(a):
    """BASELINE-size fixtures: inputs rounded to float16-representable float32 values (13 zero mantissa bits: the
    .npz compresses to about half).  Parity does not care what the input values are."""
    return np.asarray(a).astype(np.float16).astype(np.float32)


def golden_ppo(dist, learner_cls=PPO_Learner, name="ppo", size=None):
    """size=None: the small round-1 fixtures.  size="c1" / "c2": the CartPole net at the minibatch sizes of BASELINE
    configs C1 (128) and C2 (8 192); size="c4": the HalfCheetah-shape Gaussian net of configs/ppo/mujoco.yaml
    (Basic_Identical, 17-256-256-6 / 17-256-256-1, leaky_relu, tanh action activation) at its per-GPU minibatch 4 096."""
    torch.manual_seed(1)
    # the other members of the shared-trunk family Basic_MLP [128] + actor [128] + critic [128] (configs/ppo/classic_control/*.yaml,
    # box2d/{LunarLander,BipedalWalker}.yaml) at the minibatch their yaml gives: 10 envs x 256 / 8 = 320 rows
    family = {"acrobot": (6, 3), "lunar": (8, 4), "pendulum": (3, 1), "walker": (24, 4), "mountaincar": (2, 3)}
    rng = np.random.default_rng(5 if size is None else {"c1": 105, "c2": 205, "c4": 405, "acrobot": 505, "lunar": 605,
                                                       "pendulum": 705, "walker": 805, "mountaincar": 905}[size])
    act_fn = nn.LeakyReLU if (dist == "categorical" or size in ("c4",) + tuple(family)) else nn.ReLU
    init = torch.nn.init.orthogonal_
    n_updates = 3 if size is None else 2
    quant = (lambda a: a) if size is None else q16
    if size in family:
        D, A = family[size]
        bs = 320
        rep = Basic_MLP((D,), [128], None, init, act_fn, "cpu")
        actor = (CategoricalActorHead(128, [128], A, None, init, act_fn, "cpu") if dist == "categorical" else
                 GaussianActorHead(128, [128], A, None, init, act_fn, nn.Tanh, "cpu"))
        critic = ValueHead(128, [128], None, init, act_fn, "cpu")
        cfg = base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, clip_range=0.2,
                          parallels=10, end_factor_lr_decay=0.5)
    elif dist == "categorical":
        D, A, bs = 4, 2, {None: 96, "c1": 128, "c2": 8192}[size]
        rep = Basic_MLP((D,), [128], None, init, act_fn, "cpu")
        actor = CategoricalActorHead(128, [128], A, None, init, act_fn, "cpu")
        critic = ValueHead(128, [128], None, init, act_fn, "cpu")
        cfg = base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, clip_range=0.2,
                          end_factor_lr_decay=0.5)
    else:
        D, A, bs = 17, 6, {None: 80, "c4": 4096}[size]
        hid = [64, 64] if size is None else [256, 256]
        rep = Basic_Identical((D,), "cpu")
        actor = GaussianActorHead(D, hid, A, None, init, act_fn, nn.Tanh, "cpu")
        critic = ValueHead(D, hid, None, init, act_fn, "cpu")
        cfg = base_config(horizon_size=256, n_epochs=16, n_minibatch=8, vf_coef=0.25, ent_coef=0.001, clip_range=0.2,
                          gamma=0.99)
    model = SharedActorCritic(rep, actor, critic)
    # give biases / log_std non-trivial values so their gradients paths are exercised with non-zero state
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
    cb = Capture()
    if name == "a2c":
        cfg.running_steps = 40          # the LinearLR horizon of a2c_learner.py:21; short so the decay is visible
        cfg.end_factor_lr_decay = 0.5
    model64 = as_double(model)                                   # the reference ITSELF in float64 (twin below)
    learner = learner_cls(cfg, model, cb)
    batches = []
    for u in range(n_updates):
        obs = quant(np.clip(rng.standard_normal((bs, D)), -5, 5).astype(np.float32))
        with torch.no_grad():
            mo = model(torch.from_numpy(obs))
            if dist == "categorical":
                actions = rng.integers(0, A, bs).astype(np.float32)
            else:
                actions = quant(rng.standard_normal((bs, A)).astype(np.float32))
            old_logp = mo.distributions.log_prob(torch.from_numpy(actions)).numpy()
        old_logp = quant((old_logp + rng.standard_normal(bs) * 0.3).astype(np.float32))   # make ratios leave the clip range
        adv = rng.standard_normal(bs).astype(np.float32)
        adv = quant(((adv - adv.mean()) / (adv.std() + 1e-8)).astype(np.float32))
        ret = quant(rng.standard_normal(bs).astype(np.float32))
        batches.append(dict(obs=obs, actions=actions, returns=ret, advantages=adv, old_logp=old_logp,
                            values=quant(rng.standard_normal(bs).astype(np.float32))))

    def call(b, L=learner):
        return L.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                        advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]},
                        batch_size=len(b["obs"]))
    out = run_learner_updates(learner, model, cb, batches, call)
    out.update(float64_twin(learner_cls(cfg, model64, Capture()), model64, batches, lambda L, b: call(b, L)))
    if size in family:                                             # (the twin's parameters are not compared anywhere)
        out = {k: v for k, v in out.items() if "/param64/" not in k}
        out["shape"] = np.array([D, A])
    out["cfg"] = np.array([cfg.learning_rate, cfg.vf_coef, cfg.ent_coef, cfg.clip_range, cfg.grad_clip_norm,
                           getattr(cfg, "end_factor_lr_decay", 1.0),
                           cfg.running_steps if name == "a2c" else learner.total_iters])
    if size is not None:
        out["n_updates"] = np.int64(n_updates)
    np.savez_compressed(os.path.join(OUT, f"{name}_{dist}{'_' + size if size else ''}.npz"), **out)


def as_double(model):
    """model.double() of a reference policy, usable: the representation classes cast their input to float32
    (representations/mlp.py:22,56) -- for the float64 yardstick that one cast is float64; nothing else is touched."""
    from xuance.torch.rl_models.modules.outputs import RepresentationOutput
    m64 = copy.deepcopy(model).double()
    for rep in [m for m in m64.modules() if isinstance(m, (Basic_MLP, Basic_Identical))]:
        body = getattr(rep, "model", None)
        rep.forward = (lambda observations, _b=body, **kw: RepresentationOutput(
            embeddings=_b(torch.as_tensor(observations, dtype=torch.float64)) if _b is not None
            else torch.as_tensor(observations, dtype=torch.float64)))
    return m64


def float64_twin(learner64, model64, batches, call):
    """The same updates by the reference's OWN learner on `model.double()` with float64 inputs: the yardstick for quantities
    where the reference's float32 evaluation is itself further than 1e-5 (of the tensor's scale) from the exact value --
    weight gradients summed over 8 192 rows with heavy cancellation (critic.values.0.weight of the C2 fixture: the float32
    reference sits 1.5e-4 from this twin).  Stored rounded to float32 (6e-8 relative: far below what is compared).
    `call(learner, batch)` = the generator's own update call."""
    out = {}
    for u, b in enumerate(batches):
        call(learner64, {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in b.items()})
        out.update(flat(f"u{u}/grad64", {n: p.grad.detach().numpy().astype(np.float32)
                                         for n, p in model64.named_parameters() if p.grad is not None}))
        out.update(flat(f"u{u}/param64", {k: v.astype(np.float32) for k, v in sd_np(model64).items()}))
    return out


CHAIN_ROWS, CHAIN_MB, CHAIN_EPOCHS = 65536, 8, 8                 # the headline's update phase: 8 epochs x 8 minibatches of 8 192


def chain_indices(epochs=CHAIN_EPOCHS, rows=CHAIN_ROWS, n_mb=CHAIN_MB):
    """Minibatch index matrix [epochs * n_mb, rows / n_mb] of the chain fixture: epoch e visits row (a_e * i + b_e) mod rows
    (a_e odd: a permutation), cut into n_mb contiguous slices like on_policy.py:196-204 cuts its shuffled arange.  Pure integer
    arithmetic, so the test rebuilds it instead of storing 2 MB of indices (tests/test_gpu_headline.py has the same lines)."""
    i = np.arange(rows, dtype=np.int64)
    return np.stack([((2 * (1103515245 * (e + 1) % 32768) + 1) * i + 12345 * (e + 1)) % rows for e in range(epochs)]
                    ).reshape(epochs * n_mb, rows // n_mb)


def golden_ppo_chain():
    """64 CHAINED updates by the reference's PPO_Learner on the CartPole net over one synthetic 256 x 256 rollout (the headline's
    update phase: 8 epochs x 8 minibatches of 8 192 rows, advantages normalised per minibatch exactly as
    DummyOnPolicyBuffer.sample does, memory_tools.py:281-282), in float32 AND by the same learner on model.double().  The
    distance between the two after 16 and 64 updates is how far the reference's own float32 arithmetic drifts over a chain
    (PPO's clipped surrogate is discontinuous in the parameters, Adam divides by sqrt(v) + eps) -- the measured yardstick for
    the engine's chain in tests/test_gpu_headline.py.  old_logp is the initial policy's own log-probability of the stored
    action (ratio = 1 at the first update, as after a real rollout)."""
    torch.manual_seed(3)
    rng = np.random.default_rng(905)
    act_fn, init = nn.LeakyReLU, torch.nn.init.orthogonal_
    D, A, N = 4, 2, CHAIN_ROWS
    rep = Basic_MLP((D,), [128], None, init, act_fn, "cpu")
    actor = CategoricalActorHead(128, [128], A, None, init, act_fn, "cpu")
    critic = ValueHead(128, [128], None, init, act_fn, "cpu")
    cfg = base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, clip_range=0.2,
                      parallels=256, running_steps=256 * 256 * 40)
    model = SharedActorCritic(rep, actor, critic)
    model64 = as_double(model)
    learner, learner64 = PPO_Learner(cfg, model, Capture()), PPO_Learner(cfg, model64, Capture())
    obs = q16(np.clip(rng.standard_normal((N, D)), -5, 5).astype(np.float32))
    with torch.no_grad():
        dist0 = model(torch.from_numpy(obs)).distributions
        actions = dist0.stochastic_sample().numpy().astype(np.float32)
        old_logp = dist0.log_prob(torch.from_numpy(actions)).numpy().astype(np.float32)
    adv = q16((rng.standard_normal(N) * 2.0 + 0.3 * obs[:, 2]).astype(np.float32))      # raw (un-normalised) advantages
    ret = q16((5.0 + 3.0 * rng.standard_normal(N)).astype(np.float32))
    idx = chain_indices()
    out = dict(obs=obs.astype(np.float16), actions=actions.astype(np.int8), old_logp=old_logp, advantages=adv.astype(np.float16),
               returns=ret.astype(np.float16))
    out.update(flat("init", sd_np(model)))
    infos = []
    for k in range(idx.shape[0]):
        a = adv[idx[k]]
        a = (a - np.mean(a)) / (np.std(a) + 1e-8)                                        # memory_tools.py:281-282
        b = dict(obs=obs[idx[k]], actions=actions[idx[k]], returns=ret[idx[k]], advantages=a, old_logp=old_logp[idx[k]])
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], advantages=b["advantages"],
                              aux_batch={"old_logp": b["old_logp"]}, batch_size=len(a))
        infos.append([info[n] for n in ("actor_loss", "critic_loss", "entropy", "predict_value", "clip_ratio", "learning_rate")])
        a64 = adv[idx[k]].astype(np.float64)
        a64 = (a64 - np.mean(a64)) / (np.std(a64) + 1e-8)
        b64 = {n: np.asarray(v, np.float64) for n, v in b.items()}
        learner64.update(obs=b64["obs"], actions=b64["actions"], returns=b64["returns"], advantages=a64,
                         aux_batch={"old_logp": b64["old_logp"]}, batch_size=len(a))
        if k == 0:
            out.update(flat("u0/grad", {n: p.grad.numpy().copy() for n, p in model.named_parameters()}))
            out.update(flat("u0/grad64", {n: p.grad.numpy().astype(np.float32) for n, p in model64.named_parameters()}))
        if k + 1 in (1, 16, 64):
            out.update(flat(f"after{k + 1}/param", sd_np(model)))
            out.update(flat(f"after{k + 1}/param64", {n: v.astype(np.float32) for n, v in sd_np(model64).items()}))
    out["infos"] = np.asarray(infos, np.float64)
    out["cfg"] = np.array([cfg.learning_rate, cfg.vf_coef, cfg.ent_coef, cfg.clip_range, cfg.grad_clip_norm, 1.0,
                           learner.total_iters])
    out["param_names"] = np.array([n for n, _ in model.named_parameters()])
    for n in out["param_names"]:
        x32, x64 = out[f"after64/param/{n}"], out[f"after64/param64/{n}"]
        print(f"  chain drift after 64 updates {n:36s} max|f32 - f64| / max|f64| = {np.abs(x32 - x64).max() / np.abs(x64).max():.2e}")
    np.savez_compressed(os.path.join(OUT, "ppo_chain_c2.npz"), **out)


PPO_CNN_ROWS = [0, 1, 255, 511]                                  # rows of the 512 x 6400 dense weight the fixture stores


def ppo_cnn_fc_init(shape):
    """Initial values of AC_CNN_Atari's dense weight in the PPO-CNN fixture: a formula instead of 13 MB of stored numbers
    (integer arithmetic, then one IEEE multiply: any machine reproduces the bits; tests/test_gpu_ppo.py has the same lines)."""
    i, j = np.meshgrid(np.arange(shape[0], dtype=np.int64), np.arange(shape[1], dtype=np.int64), indexing="ij")
    return (((i * 131 + j * 7919 + 17) % 2003 - 1001).astype(np.float64) * 3e-5).astype(np.float32)


def golden_ppo_cnn():
    """PPO_Learner on SharedActorCritic(AC_CNN_Atari, CategoricalActorHead([]), ValueHead([])) -- configs/ppo/atari.yaml:
    84 x 84 x 4 uint8 frame stacks, conv 32/64/64 (k 8/4/3, s 4/2/1), Flatten, Linear(6400, 512), heads on the embedding -- two
    updates on 32 frames each.  3.36 M parameters, 3.28 M of them the dense weight: that tensor is INITIALISED from a formula
    (ppo_cnn_fc_init) and only rows PPO_CNN_ROWS of its gradient / values / Adam moments are stored (`rows/<name>`); every other
    tensor is stored whole.  Float64 twin of the gradients as in the other PPO fixtures."""
    from xuance.torch.rl_models.representations.cnn import AC_CNN_Atari
    torch.manual_seed(11)
    rng = np.random.default_rng(611)
    init = torch.nn.init.orthogonal_
    A, bs = 4, 32
    rep = AC_CNN_Atari((84, 84, 4), [8, 4, 3], [4, 2, 1], [32, 64, 64], None, init, nn.ReLU, "cpu", fc_hidden_sizes=[512])
    actor = CategoricalActorHead(512, [], A, None, init, nn.ReLU, "cpu")
    critic = ValueHead(512, [], None, init, nn.ReLU, "cpu")
    model = SharedActorCritic(rep, actor, critic)
    fc = "representation.model.7.weight"
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
            if n == fc:
                p.copy_(torch.from_numpy(ppo_cnn_fc_init(tuple(p.shape))))
    cfg = base_config(horizon_size=128, n_epochs=4, n_minibatch=4, parallels=8, running_steps=128 * 8 * 100, learning_rate=2.5e-4,
                      vf_coef=0.25, ent_coef=0.01, clip_range=0.2, gamma=0.99, end_factor_lr_decay=0.5)
    model64 = copy.deepcopy(model).double()
    for m_ in [m for m in model64.modules() if isinstance(m, AC_CNN_Atari)]:     # its forward casts to float32 (cnn.py:100)
        body = m_.model
        from xuance.torch.rl_models.modules.outputs import RepresentationOutput
        m_.forward = lambda observations, _b=body, **kw: RepresentationOutput(
            embeddings=_b((torch.as_tensor(observations, dtype=torch.float64) / 255.0).permute((0, 3, 1, 2))))
    cb = Capture()
    learner, learner64 = PPO_Learner(cfg, model, cb), PPO_Learner(cfg, model64, Capture())
    batches = []
    for u in range(2):
        obs = (rng.integers(0, 16, (bs, 84, 84, 4)) * 17).astype(np.uint8)
        with torch.no_grad():
            actions = rng.integers(0, A, bs).astype(np.float32)
            old_logp = model(obs).distributions.log_prob(torch.from_numpy(actions)).numpy()
        old_logp = (old_logp + rng.standard_normal(bs) * 0.3).astype(np.float32)
        adv = rng.standard_normal(bs).astype(np.float32)
        adv = ((adv - adv.mean()) / (adv.std() + 1e-8)).astype(np.float32)
        batches.append(dict(obs=obs, actions=actions, returns=rng.standard_normal(bs).astype(np.float32), advantages=adv,
                            old_logp=old_logp, values=rng.standard_normal(bs).astype(np.float32)))

    def call(b, L=learner):
        return L.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"], advantages=b["advantages"],
                        aux_batch={"old_logp": b["old_logp"]}, batch_size=bs)
    out = run_learner_updates(learner, model, cb, batches, call)
    out.update(float64_twin(learner64, model64, batches, lambda L, b: call(
        {k: (v if k == "obs" else v) for k, v in b.items()}, L)))
    rows = np.asarray(PPO_CNN_ROWS)
    for k in list(out):
        if k.endswith("/" + fc):
            if k.startswith("init/"):
                assert np.array_equal(out[k], ppo_cnn_fc_init(out[k].shape))
                del out[k]                                       # rebuilt from the formula
            else:
                out[k] = out[k][rows]
    out["rows/" + fc] = rows
    out["cfg"] = np.array([cfg.learning_rate, cfg.vf_coef, cfg.ent_coef, cfg.clip_range, cfg.grad_clip_norm, cfg.end_factor_lr_decay,
                           learner.total_iters])
    out["n_updates"] = np.int64(2)
    np.savez_compressed(os.path.join(OUT, "ppo_cnn_atari.npz"), **out)


def golden_ppokl(dist):
    """PPOKL_Learner (ppokl_learner.py:14-101).  The reference's update reads `model_output.distribution` (:48) while
    ActorCriticOutput names the field `distributions`, so it raises AttributeError as shipped.  The fixture runs the
    UNMODIFIED learner on a model whose forward output also carries that attribute name (an alias set on the output object:
    nothing of the learner is restated or patched); old_dists are reference distribution objects produced by
    split_distributions, as PPOKL_Agent stores them (ppokl_agent.py:20-30)."""
    from xuance.torch.learners import PPOKL_Learner
    from xuance.torch.rl_models.modules.distributions import (CategoricalDistribution, DiagGaussianDistribution,
                                                               split_distributions)
    torch.manual_seed(3)
    rng = np.random.default_rng(31)
    init = torch.nn.init.orthogonal_
    if dist == "categorical":
        D, A, bs, act_fn = 4, 3, 96, nn.LeakyReLU
        rep = Basic_MLP((D,), [128], None, init, act_fn, "cpu")
        actor = CategoricalActorHead(128, [128], A, None, init, act_fn, "cpu")
        critic = ValueHead(128, [128], None, init, act_fn, "cpu")
    else:
        D, A, bs, act_fn = 17, 6, 80, nn.LeakyReLU
        rep = Basic_Identical((D,), "cpu")
        actor = GaussianActorHead(D, [64, 64], A, None, init, act_fn, nn.Tanh, "cpu")
        critic = ValueHead(D, [64, 64], None, init, act_fn, "cpu")
    cfg = base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, target_kl=0.02, kl_coef=1.0,
                      end_factor_lr_decay=0.5)
    model = SharedActorCritic(rep, actor, critic)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
    inner = model.forward

    def forward_with_alias(*a, **k):
        out = inner(*a, **k)
        out.distribution = out.distributions
        return out
    model.forward = forward_with_alias
    cb = Capture()
    learner = PPOKL_Learner(cfg, model, cb)
    coefs, batches = [], []
    for u in range(4):
        obs = np.clip(rng.standard_normal((bs, D)), -5, 5).astype(np.float32)
        noise = ([0.05, 0.6, 0.02, 0.3] if dist == "categorical" else [0.05, 0.3, 0.02, 0.2])[u]   # small / large KL: the coefficient halves, doubles, ...
        with torch.no_grad():
            d_new = inner(torch.from_numpy(obs)).distributions
            if dist == "categorical":
                old = CategoricalDistribution(A)
                old.set_param(logits=d_new.logits + noise * torch.from_numpy(rng.standard_normal((bs, A)).astype(np.float32)))
                actions = rng.integers(0, A, bs).astype(np.float32)
                old_a, old_b = old.logits.numpy().copy(), np.zeros((bs, A), np.float32)
            else:
                old = DiagGaussianDistribution(A)
                # (one std vector for the whole batch: GaussianActorHead's log_std is a parameter, actor_head.py:64-71)
                # actions as a rollout would draw them (near mu, in units of std), the old policy a small step away
                old.set_param(d_new.mu + noise * d_new.std * torch.from_numpy(rng.standard_normal((bs, A)).astype(np.float32)),
                              d_new.std * torch.from_numpy(np.exp(0.3 * noise * rng.standard_normal(A)).astype(np.float32)))
                actions = (d_new.mu + d_new.std * torch.from_numpy(rng.standard_normal((bs, A)).astype(np.float32))).numpy().copy()
                old_a, old_b = old.mu.numpy().copy(), np.broadcast_to(old.std.numpy(), (bs, A)).copy()
        adv = rng.standard_normal(bs).astype(np.float32)
        adv = ((adv - adv.mean()) / (adv.std() + 1e-8)).astype(np.float32)
        batches.append(dict(obs=obs, actions=actions, returns=rng.standard_normal(bs).astype(np.float32), advantages=adv,
                            old_a=old_a, old_b=old_b, _old=split_distributions(old)))

    def call(b):
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], advantages=b["advantages"],
                              aux_batch={"old_dist": b["_old"]}, batch_size=len(b["obs"]))
        coefs.append(float(learner.kl_coef))
        return info
    clean = [{k: v for k, v in b.items() if not k.startswith("_")} for b in batches]
    out = run_learner_updates(learner, model, cb, batches, call)
    out = {k: v for k, v in out.items() if "/batch/_old" not in k}
    for u, b in enumerate(clean):
        out.update(flat(f"u{u}/batch", b))
    out["kl_coef_after"] = np.array(coefs, np.float64)
    out["cfg"] = np.array([cfg.learning_rate, cfg.vf_coef, cfg.ent_coef, cfg.target_kl, cfg.kl_coef, cfg.grad_clip_norm,
                           cfg.end_factor_lr_decay, learner.total_iters])
    out["n_updates"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, f"ppokl_{dist}.npz"), **out)


# ------------------------------------------------------------------------------ DQN
def golden_dqn(kind, learner_cls=None, name=None, model_cls=None, size=None, levels=None, huber=None):
    """learner_cls: DQN_Learner (default), DDQN_Learner (ddqn_learner.py:39-47, the double-Q target) or DuelDQN_Learner
    with model_cls=DuelingDeepQNetwork (dueldqn_learner.py:28-75, q_head.py:42-80).
    huber: the UNMODIFIED learner's update with its loss module (the attribute `mse_loss`, dqn_learner.py:25,46) set to
    nn.HuberLoss(delta=huber) -- the loss the reference's learners build behind `use_huber_loss` / `huber_delta`
    (learners/base/marl_learner.py:193-197; there with reduction "none" + a masked mean, here the module's own "mean", which
    is what DQN_Learner's call site `self.mse_loss(predictQ, targetQ)` expects).  Rewards are scaled by 2 so that TD errors
    fall on both sides of delta.
    size="c3" (kind "cnn"): the batch of configs/dqn/atari.yaml:27 (32 frames of 84x84x4), two updates, lr 1e-4, no clip;
    frames take 16 grey levels so the file compresses."""
    learner_cls = learner_cls or DQN_Learner
    model_cls = model_cls or DeepQNetwork
    torch.manual_seed(2)
    rng = np.random.default_rng(9 if size is None else 309)
    n_updates = 3 if size is None else 2
    init = torch.nn.init.orthogonal_
    if kind == "mlp":
        D, A, bs = 6, 4, 32
        rep = Basic_MLP((D,), [64], None, init, nn.ReLU, "cpu")
        hidden = [64]
    else:
        A, bs = 4, (4 if size is None else 32)
        rep = Basic_CNN((84, 84, 4), [8, 4, 3], [4, 2, 1], [32, 64, 64], None, init, nn.ReLU, "cpu")
        hidden = [512]
    model = model_cls(rep, hidden, sp.Discrete(A), None, init, nn.ReLU, "cpu")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
        # de-synchronise target from eval so the target path is really exercised
        for n, p in model.named_parameters():
            if n.startswith("target_"):
                p.add_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.05))
    cfg = base_config(learning_rate=1e-3 if size is None else 1e-4, gamma=0.99, sync_frequency=2, start_training=0,
                      training_frequency=1, use_grad_clip=(kind == "mlp"), grad_clip_norm=0.5)
    cb = Capture()
    learner = learner_cls(cfg, model, cb)
    if huber is not None:
        learner.mse_loss = nn.HuberLoss(reduction="mean", delta=float(huber))
    batches = []
    for u in range(n_updates):
        if kind == "mlp":
            obs = rng.standard_normal((bs, D)).astype(np.float32)
            nxt = rng.standard_normal((bs, D)).astype(np.float32)
        elif levels is not None:                       # (a few grey levels: the file compresses)
            obs = (rng.integers(0, levels, (bs, 84, 84, 4)) * (255 // (levels - 1))).astype(np.uint8)
            nxt = (rng.integers(0, levels, (bs, 84, 84, 4)) * (255 // (levels - 1))).astype(np.uint8)
        elif size is not None:
            obs = (rng.integers(0, 16, (bs, 84, 84, 4)) * 17).astype(np.uint8)
            nxt = (rng.integers(0, 16, (bs, 84, 84, 4)) * 17).astype(np.uint8)
        else:
            obs = rng.integers(0, 256, (bs, 84, 84, 4)).astype(np.uint8)
            nxt = rng.integers(0, 256, (bs, 84, 84, 4)).astype(np.uint8)
        batches.append(dict(obs=obs, obs_next=nxt, actions=rng.integers(0, A, bs).astype(np.float32),
                            rewards=(rng.standard_normal(bs) * (2.0 if huber is not None else 1.0)).astype(np.float32),
                            terminals=(rng.random(bs) < 0.2).astype(np.float32)))

    def call(b):
        return learner.update(batch_size=bs, **b)
    out = run_learner_updates(learner, model, cb, batches, call)
    out["cfg"] = np.array([cfg.learning_rate, cfg.gamma, cfg.sync_frequency, cfg.grad_clip_norm,
                           float(cfg.use_grad_clip), learner.total_iters])
    if huber is not None:
        out["huber_delta"] = np.float64(huber)
        td = np.concatenate([out[f"u{u}/cb/predictQ"] - out[f"u{u}/cb/targetQ"] for u in range(n_updates)])
        assert (np.abs(td) > huber).any() and (np.abs(td) < huber).any(), "both branches of the Huber loss must be taken"
    if size is not None:
        out["n_updates"] = np.int64(n_updates)
    np.savez_compressed(os.path.join(OUT, f"{name or 'dqn'}_{kind}{'_' + size if size else ''}.npz"), **out)


# ------------------------------------------------------------------------------ QMIX (feed-forward)
def golden_qmix(double_q, algo="qmix", size=None):
    """algo: qmix (QMIX_Learner + QMIX_Mixer), vdn (VDN_Learner + VDN_Mixer, vdn_learner.py:13-106), iql (IQL_Learner +
    IndependentMixer, iql_learner.py:13-142): same feed-forward agents, same batches.
    size="c5": batch 32 (configs/qmix/sc2/3m.yaml:32)."""
    from xuance.torch.rl_models.critics.base_critics import DiscreteActionValueCritic
    from xuance.torch.rl_models.representations.agent_feature import AgentFeatureEncoder
    torch.manual_seed(3)
    rng = np.random.default_rng(13 if size is None else 513)
    N, O, S, A, B = 3, 30, 48, 9, (16 if size is None else 32)
    agent_keys = [f"agent_{i}" for i in range(N)]
    grouping = AgentGrouping.shared(agent_keys)            # agents_marl.py:210-215
    group = grouping.group_keys[0]
    init = torch.nn.init.orthogonal_
    obs_rep = Basic_MLP((O,), [64], None, init, nn.ReLU, "cpu")
    from xuance.torch.rl_models.modules.identity_encoder import build_identity_encoder, IdentityFeatureFusion
    ident = build_identity_encoder(num_identities=N, mode="none", embedding_dim=None, device="cpu")   # agents_marl.py:313-318
    fusion = IdentityFeatureFusion(observation_feature_dim=64, identity_feature_dim=ident.output_dim, mode="concat")
    rep = AgentFeatureEncoder(representation=obs_rep, identity_encoder=ident, fusion=fusion)
    critic = DiscreteActionValueCritic(representation=rep, action_space=sp.Discrete(A), critic_hidden_size=[64],
                                       normalizer=None, initializer=init, activation=nn.ReLU, device="cpu")
    from xuance.torch.rl_models.heads import VDN_Mixer, IndependentMixer
    from xuance.torch.learners import VDN_Learner, IQL_Learner
    mixer = {"qmix": lambda: QMIX_Mixer(S, 32, 32, N, "cpu"), "vdn": VDN_Mixer, "iql": IndependentMixer}[algo]()
    model = MixingQNetwork(grouping, nn.ModuleDict({group: critic}), mixer, use_rnn=False, device="cpu")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("target_"):
                p.add_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.05))
    cfg = base_config(learning_rate=7e-4, gamma=0.99, sync_frequency=2, start_training=0, training_frequency=1,
                      use_parameter_sharing=True, double_q=double_q, use_actions_mask=True, use_rnn=False,
                      n_epochs=8, grad_clip_norm=10.0)
    cb = Capture()
    learner = {"qmix": QMIX_Learner, "vdn": VDN_Learner, "iql": IQL_Learner}[algo](cfg, grouping, model, cb)
    batches, samples = [], []
    for u in range(3):
        avail = (rng.random((B, N, A)) < 0.7)
        avail[..., 0] = True
        avail_n = (rng.random((B, N, A)) < 0.7)
        avail_n[..., 0] = True
        acts = np.zeros((B, N), np.float32)
        for b in range(B):
            for i in range(N):
                acts[b, i] = rng.choice(np.flatnonzero(avail[b, i]))
        term = rng.random((B, N)) < 0.5
        term[: B // 4] = True
        b = dict(obs=rng.standard_normal((B, N, O)).astype(np.float32),
                 obs_next=rng.standard_normal((B, N, O)).astype(np.float32), actions=acts,
                 rewards=rng.standard_normal((B, N)).astype(np.float32), terminals=term,
                 agent_mask=(rng.random((B, N)) < 0.85), avail_actions=avail, avail_actions_next=avail_n,
                 state=rng.standard_normal((B, S)).astype(np.float32),
                 state_next=rng.standard_normal((B, S)).astype(np.float32))
        batches.append(b)
        sample = {k: {a: b[k][:, i] for i, a in enumerate(agent_keys)}
                  for k in ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions",
                            "avail_actions_next")}
        sample.update(state=b["state"], state_next=b["state_next"], batch_size=B)
        samples.append(sample)
    it = iter(samples)
    out = run_learner_updates(learner, model, cb, batches, lambda b: learner.update(next(it)))
    out["cfg"] = np.array([cfg.learning_rate, cfg.gamma, cfg.sync_frequency, cfg.grad_clip_norm, float(double_q),
                           learner.total_iters])
    out["group"] = np.array(group)
    np.savez_compressed(os.path.join(OUT, f"{algo}_ff_{'double' if double_q else 'single'}{'_' + size if size else ''}.npz"), **out)


def golden_qmix_rnn(double_q=True, fixed=False, rnn="GRU", size=None):
    """QMIX_Learner.update with recurrent agents (3m.yaml defaults: Basic_RNN fc 64 + GRU 64, q_hidden 64) on episode
    samples in the layout MARL_OffPolicyBuffer_RNN.sample returns (memory_tools_marl.py:970-996).

    fixed=False: the UNMODIFIED reference, use_actions_mask off.  Two properties of the unmodified recurrent branch:
      (1) with use_actions_mask on it cannot run at all: iql_learner.py:78 slices the AGENT axis
          (`avail_actions.group(group)[:, 1:]` on a [B, N, T+1, A] tensor) and line 81 raises IndexError
          (mask [B, N-1, T+1, A] vs values [B, N, T, A]);
      (2) iql_learner.py:58 re-slices q_eval (`v[:, :, :-1]`) INSIDE `with torch.no_grad()` (:49), so q_eval reaches the
          loss detached: the agent networks (fc, GRU, Q head) receive no gradient, only the mixer trains.
    fixed=True runs a subclass that restates _forward_transitions with exactly those two lines changed (time-axis
    slice `[:, :, 1:]`; q_eval sliced outside no_grad) -- the algorithm the surrounding code implies.  The fixture
    name says so (`..._fixed`); it pins the full back-propagation-through-time path."""
    from xuance.torch.rl_models.critics.base_critics import DiscreteActionValueCritic
    from xuance.torch.rl_models.representations.agent_feature import AgentFeatureEncoder
    from xuance.torch.rl_models.representations import Basic_RNN
    from xuance.torch.rl_models.modules.identity_encoder import build_identity_encoder, IdentityFeatureFusion
    torch.manual_seed(4)
    rng = np.random.default_rng(17 if size is None else 517)
    N, O, S, A, B, T = (3, 30, 48, 9, 8, 12) if size is None else (3, 30, 48, 9, 32, 60)   # c5: 3m.yaml:32, smac.rst:19
    n_updates = 3 if size is None else 2
    quant = (lambda a: a) if size is None else q16
    agent_keys = [f"agent_{i}" for i in range(N)]
    grouping = AgentGrouping.shared(agent_keys)
    group = grouping.group_keys[0]
    init = torch.nn.init.orthogonal_
    obs_rep = Basic_RNN((O,), None, None, init, nn.ReLU, "cpu", fc_hidden_sizes=[64], recurrent_hidden_size=64,
                        N_recurrent_layers=1, dropout=0, rnn=rnn)
    ident = build_identity_encoder(num_identities=N, mode="none", embedding_dim=None, device="cpu")
    fusion = IdentityFeatureFusion(observation_feature_dim=64, identity_feature_dim=ident.output_dim, mode="concat")
    rep = AgentFeatureEncoder(representation=obs_rep, identity_encoder=ident, fusion=fusion)
    critic = DiscreteActionValueCritic(representation=rep, action_space=sp.Discrete(A), critic_hidden_size=[64],
                                       normalizer=None, initializer=init, activation=nn.ReLU, device="cpu")
    mixer = QMIX_Mixer(S, 32, 32, N, "cpu")
    model = MixingQNetwork(grouping, nn.ModuleDict({group: critic}), mixer, use_rnn=True, device="cpu")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("target_"):
                p.add_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.05))
            elif "bias" in n and "rnn" in n:            # GRU biases are initialised to 0: make their paths non-trivial
                p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
    cfg = base_config(learning_rate=7e-4, gamma=0.99, sync_frequency=2, start_training=0, training_frequency=1,
                      use_parameter_sharing=True, double_q=double_q, use_actions_mask=fixed, use_rnn=True,
                      n_epochs=8, grad_clip_norm=10.0, episode_length=T, parallels=4, running_steps=4800)
    cb = Capture()
    cls = QMIX_Learner
    if fixed:
        class QMIX_Learner_Fixed(QMIX_Learner):
            def _forward_transitions(self, batch):                # iql_learner.py:37-83, recurrent branch, two lines changed
                rnn_states = self.model.init_rnn_states(batch.batch_size)
                out = self.model(observations=batch.observations, agent_indices=batch.agent_indices,
                                 avail_actions=batch.avail_actions, rnn_states=rnn_states)
                q_eval = out.values
                q_eval.grouped_tensor = {k: v[:, :, :-1] for k, v in q_eval.grouped_tensor.items()}     # outside no_grad
                with torch.no_grad():
                    actions_next = out.actions
                    q_next = self.model.Qtarget(observations=batch.observations, agent_indices=batch.agent_indices,
                                                rnn_states=rnn_states).values
                    q_next.grouped_tensor = {k: v[:, :, 1:] for k, v in q_next.grouped_tensor.items()}
                    actions_next.grouped_tensor = {k: v[:, :, 1:] for k, v in actions_next.grouped_tensor.items()}
                for g_ in self.group_keys:
                    q_next.group(g_)[batch.avail_actions.group(g_)[:, :, 1:] == 0] = -1e10                # time axis
                return q_eval, q_next, actions_next
        cls = QMIX_Learner_Fixed
    learner = cls(cfg, grouping, model, cb)
    batches, samples = [], []
    for u in range(n_updates):
        avail = (rng.random((B, N, T + 1, A)) < 0.7)
        avail[..., 0] = True
        acts = np.zeros((B, N, T), np.float32)
        for b in range(B):
            for i in range(N):
                for t in range(T):
                    acts[b, i, t] = rng.choice(np.flatnonzero(avail[b, i, t]))
        lengths = rng.integers(3, T + 1, B)
        lengths[0] = T
        filled = np.arange(T)[None, :] < lengths[:, None]
        term = np.zeros((B, N, T), bool)
        for b in range(B):
            if b % 2 == 0:
                term[b, :, lengths[b] - 1] = True               # episode ended by termination (all agents)
            else:
                term[b, 0, lengths[b] - 1] = True               # only one agent: terminals_tot stays 0 (truncation)
        b = dict(obs=quant(rng.standard_normal((B, N, T + 1, O)).astype(np.float32)), actions=acts,
                 rewards=quant(rng.standard_normal((B, N, T)).astype(np.float32)), terminals=term,
                 agent_mask=(rng.random((B, N, T)) < 0.85), avail_actions=avail,
                 state=quant(rng.standard_normal((B, T + 1, S)).astype(np.float32)), filled=filled)
        batches.append(b)
        sample = {k: {a: b[k][:, i] for i, a in enumerate(agent_keys)}
                  for k in ("obs", "actions", "rewards", "terminals", "agent_mask", "avail_actions")}
        sample.update(state=b["state"], filled=b["filled"], batch_size=B, sequence_length=T)
        samples.append(sample)
    it = iter(samples)
    out = run_learner_updates(learner, model, cb, batches, lambda b: learner.update(next(it)))
    out["cfg"] = np.array([cfg.learning_rate, cfg.gamma, cfg.sync_frequency, cfg.grad_clip_norm, float(double_q),
                           learner.total_iters])
    out["group"] = np.array(group)
    if size is not None:
        out["n_updates"] = np.int64(n_updates)
    np.savez_compressed(os.path.join(OUT, f"qmix_{'rnn' if rnn == 'GRU' else rnn.lower()}_{'double' if double_q else 'single'}{'_fixed' if fixed else ''}{'_' + size if size else ''}.npz"),
                        **out)


def golden_marl_rnn_buffer():
    """MARL_OffPolicyBuffer_RNN (memory_tools_marl.py:770-996): store / finish_path / clear_episodes / sample on a small
    scripted scenario -- ragged episode lengths, several envs finishing in the same step, the ring wrapping, and short
    episodes that follow longer ones in the same staging row (store_episodes copies the stale tail, :935-949)."""
    from xuance.common.memory_tools_marl import MARL_OffPolicyBuffer_RNN
    rng = np.random.default_rng(23)
    n_envs, N, O, A, S, T, cap, bs = 3, 2, 3, 4, 5, 5, 6, 6
    keys = [f"agent_{i}" for i in range(N)]
    buf = MARL_OffPolicyBuffer_RNN(agent_keys=keys, state_space=sp.Box(-np.inf, np.inf, (S,), np.float32),
                                   obs_space={k: sp.Box(-np.inf, np.inf, (O,), np.float32) for k in keys},
                                   act_space={k: sp.Discrete(A) for k in keys}, n_envs=n_envs, buffer_size=cap,
                                   batch_size=bs, max_episode_steps=T, use_actions_mask=True,
                                   avail_actions_shape={k: (A,) for k in keys})
    lengths = [[5, 2, 4, 1], [3, 5, 2, 2], [1, 1, 5, 3]]           # per env: successive episode lengths
    ep, es = [0] * n_envs, np.zeros(n_envs, np.int64)
    out, n_steps = {}, 12
    for t in range(n_steps):
        if t == 8:
            buf.clear_episodes()                                    # what run_episodes does when it is entered again (:459)
            es[:] = 0
            for e in range(n_envs):
                ep[e] += 1
        d = dict(obs=rng.standard_normal((n_envs, N, O)).astype(np.float32),
                 actions=rng.integers(0, A, (n_envs, N)).astype(np.float32),
                 rewards=rng.standard_normal((n_envs, N)).astype(np.float32),
                 terminals=rng.random((n_envs, N)) < 0.3, agent_mask=rng.random((n_envs, N)) < 0.8,
                 avail_actions=rng.random((n_envs, N, A)) < 0.7, state=rng.standard_normal((n_envs, S)).astype(np.float32),
                 term_obs=rng.standard_normal((n_envs, N, O)).astype(np.float32),
                 term_state=rng.standard_normal((n_envs, S)).astype(np.float32),
                 term_avail=rng.random((n_envs, N, A)) < 0.7)
        step = {k: {a: d[k][:, i] for i, a in enumerate(keys)} for k in ("obs", "actions", "rewards", "terminals",
                                                                           "agent_mask", "avail_actions")}
        step.update(state=d["state"], episode_steps=es.copy())
        buf.store(**step)
        done = np.zeros(n_envs, bool)
        for e in range(n_envs):
            if es[e] + 1 >= lengths[e][ep[e] % 4]:
                done[e] = True
                buf.finish_path(e, obs={a: d["term_obs"][e, i] for i, a in enumerate(keys)}, state=d["term_state"][e],
                                avail_actions={a: d["term_avail"][e, i] for i, a in enumerate(keys)},
                                episode_step=int(es[e] + 1))
        d["episode_steps"], d["done"] = es.copy(), done
        out.update(flat(f"t{t}", d))
        for e in range(n_envs):
            if done[e]:
                es[e], ep[e] = 0, ep[e] + 1
            else:
                es[e] += 1
        out[f"t{t}/ptr_size"] = np.array([buf.ptr, buf.size])
    for k, v in buf.data.items():
        out[f"data/{k}"] = np.stack([v[a] for a in keys], 2) if isinstance(v, dict) else v   # [cap, slots, N, ...]
    np.random.seed(7)
    smp = buf.sample()
    np.random.seed(7)
    out["sample/idx"] = np.random.choice(buf.size, bs)
    for k, v in smp.items():
        if isinstance(v, dict):
            out[f"sample/{k}"] = np.stack([v[a] for a in keys], 1)                              # [B, N, slots, ...]
        elif isinstance(v, np.ndarray):
            out[f"sample/{k}"] = v
    out["meta"] = np.array([n_envs, N, O, A, S, T, cap, bs, n_steps])
    np.savez_compressed(os.path.join(OUT, "marl_rnn_buffer.npz"), **out)


def golden_marl_ff_buffer():
    """MARL_OffPolicyBuffer (memory_tools_marl.py:634-767): store over a wrapping ring (bool fields included), then
    sample() with the two NumPy global-RNG draws (:755-756).  tests/golden/marl_ff_buffer.npz."""
    from xuance.common.memory_tools_marl import MARL_OffPolicyBuffer
    rng = np.random.default_rng(29)
    n_envs, n_size, N, O, A, S, bs, n_steps = 4, 6, 3, 5, 4, 7, 10, 9
    keys = [f"agent_{i}" for i in range(N)]
    buf = MARL_OffPolicyBuffer(agent_keys=keys, state_space=sp.Box(-np.inf, np.inf, (S,), np.float32),
                               obs_space={k: sp.Box(-np.inf, np.inf, (O,), np.float32) for k in keys},
                               act_space={k: sp.Discrete(A) for k in keys}, n_envs=n_envs, buffer_size=n_envs * n_size,
                               batch_size=bs, use_actions_mask=True, avail_actions_shape={k: (A,) for k in keys})
    out = {}
    per_agent = ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions", "avail_actions_next")
    for t in range(n_steps):
        d = dict(obs=rng.standard_normal((n_envs, N, O)).astype(np.float32),
                 obs_next=rng.standard_normal((n_envs, N, O)).astype(np.float32),
                 actions=rng.integers(0, A, (n_envs, N)).astype(np.float32),
                 rewards=rng.standard_normal((n_envs, N)).astype(np.float32),
                 terminals=rng.random((n_envs, N)) < 0.2, agent_mask=rng.random((n_envs, N)) < 0.9,
                 avail_actions=rng.random((n_envs, N, A)) < 0.7, avail_actions_next=rng.random((n_envs, N, A)) < 0.7,
                 state=rng.standard_normal((n_envs, S)).astype(np.float32),
                 state_next=rng.standard_normal((n_envs, S)).astype(np.float32))
        step = {k: {a: d[k][:, i] for i, a in enumerate(keys)} for k in per_agent}
        step.update(state=d["state"], state_next=d["state_next"])
        buf.store(**step)
        out.update(flat(f"t{t}", d))
        out[f"t{t}/ptr_size"] = np.array([buf.ptr, buf.size])
    for k, v in buf.data.items():
        out[f"data/{k}"] = np.stack([v[a] for a in keys], 2) if isinstance(v, dict) else v      # [n_envs, n_size, N, ...]
    np.random.seed(11)
    smp = buf.sample()
    np.random.seed(11)
    out["sample/env"], out["sample/step"] = np.random.choice(n_envs, bs), np.random.choice(buf.size, bs)
    for k, v in smp.items():
        if isinstance(v, dict):
            out[f"sample/{k}"] = np.stack([v[a] for a in keys], 1)                                # [B, N, ...]
        elif isinstance(v, np.ndarray):
            out[f"sample/{k}"] = v
    out["meta"] = np.array([n_envs, n_size, N, O, A, S, bs, n_steps])
    np.savez_compressed(os.path.join(OUT, "marl_ff_buffer.npz"), **out)


def golden_checkpoint(decay=False):
    """Checkpoint compatibility: a `.pth` written by the reference's Learner.save_model
    (drl_learner.py:64-93) after two PPO updates, and the parameters the REFERENCE reaches when a fresh learner loads that
    file (load_model, :95-157) and makes a third update.  tests/golden/ppo_ckpt_ref.pth + ppo_ckpt.npz.
    decay=True: end_factor_lr_decay 0.5 over a short LinearLR horizon (128 iterations), so the schedule is visible: the
    reference resumes from the DECAYED param-group lr with a fresh scheduler (its chained LinearLR form then continues as
    lr_saved * (1 + (ef - 1) * j / total), j = steps since the resume).  ppo_ckpt_decay_ref.pth + ppo_ckpt_decay.npz."""
    import copy
    torch.manual_seed(6)
    rng = np.random.default_rng(31 if not decay else 131)
    init = torch.nn.init.orthogonal_
    D, A, bs = 4, 2, 64

    def build():
        rep = Basic_MLP((D,), [128], None, init, nn.LeakyReLU, "cpu")
        actor = CategoricalActorHead(128, [128], A, None, init, nn.LeakyReLU, "cpu")
        critic = ValueHead(128, [128], None, init, nn.LeakyReLU, "cpu")
        m = SharedActorCritic(rep, actor, critic)
        cfg = base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, clip_range=0.2)
        if decay:
            cfg.running_steps, cfg.end_factor_lr_decay = 2048, 0.5
        return m, PPO_Learner(cfg, m, Capture())
    model, learner = build()
    batches = []
    for u in range(3):
        obs = np.clip(rng.standard_normal((bs, D)), -5, 5).astype(np.float32)
        actions = rng.integers(0, A, bs).astype(np.float32)
        with torch.no_grad():
            old_logp = model(torch.from_numpy(obs)).distributions.log_prob(torch.from_numpy(actions)).numpy()
        adv = rng.standard_normal(bs).astype(np.float32)
        batches.append(dict(obs=obs, actions=actions, returns=rng.standard_normal(bs).astype(np.float32), advantages=adv,
                            old_logp=(old_logp + rng.standard_normal(bs) * 0.2).astype(np.float32),
                            values=rng.standard_normal(bs).astype(np.float32)))

    def call(l, b):
        return l.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                        advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]}, batch_size=bs)
    out = flat("init", sd_np(model))
    call(learner, batches[0]); info1 = call(learner, batches[1])
    tag = "ppo_ckpt_decay" if decay else "ppo_ckpt"
    path = os.path.join(OUT, tag + "_ref.pth")
    learner.save_model(path)
    out.update(flat("saved", sd_np(model)))
    model2, learner2 = build()                                      # fresh process: new model, new optimiser
    learner2.load_model(path)
    info = call(learner2, batches[2])
    out.update(flat("resumed", sd_np(model2)))
    for u, b in enumerate(batches):
        out.update(flat(f"u{u}/batch", b))
    out["resumed_info/learning_rate"] = np.float64(info["learning_rate"])
    out["resumed_info/actor_loss"] = np.float64(info["actor_loss"])
    if decay:
        out["saved_info/learning_rate"] = np.float64(info1["learning_rate"])
        out["total_iters"] = np.int64(learner.total_iters)
        info4 = call(learner2, batches[0])                            # a second update after the resume
        out["resumed2_info/learning_rate"] = np.float64(info4["learning_rate"])
        out.update(flat("resumed2", sd_np(model2)))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)


def golden_per_buffer():
    """PerOffPolicyBuffer (memory_tools.py:471-598, segtree_tool.py): stores past the wrap, stratified proportional sampling
    with recorded `random.random()` uniforms, importance weights, priority updates (zeros, duplicates, new maxima) -- with
    float64 priorities, i.e. the arithmetic of the NumPy < 2 the reference pins (see HipPerOffPolicyBuffer)."""
    import random
    from xuance.common.memory_tools import PerOffPolicyBuffer
    rng = np.random.default_rng(41)
    n_envs, n_size, D, bs, alpha = 4, 6, 3, 8, 0.6
    k = bs // n_envs
    buf = PerOffPolicyBuffer(sp.Box(-np.inf, np.inf, (D,), np.float32), sp.Discrete(3), None, n_envs, n_envs * n_size, bs, alpha)
    out, ev = {}, 0

    def store():
        nonlocal ev
        d = dict(obs=rng.standard_normal((n_envs, D)).astype(np.float32), acts=rng.integers(0, 3, n_envs).astype(np.float32),
                 rews=rng.standard_normal(n_envs).astype(np.float32), terminals=(rng.random(n_envs) < 0.2),
                 next_obs=rng.standard_normal((n_envs, D)).astype(np.float32))
        buf.store(d["obs"], d["acts"], d["rews"], d["terminals"], d["next_obs"])
        out.update(flat(f"e{ev}/store", d)); ev += 1

    def sample_update(beta):
        nonlocal ev
        random.seed(100 + ev)
        uni = np.array([[random.random() for _ in range(k)] for _ in range(n_envs)])
        random.seed(100 + ev)
        smp = buf.sample(beta)
        pr = np.abs(rng.standard_normal((n_envs, k))).astype(np.float32).astype(np.float64) * 2.0
        pr[0, 0] = 0.0                                              # the `priority == 0 -> 1e-8` branch (:590-591)
        pr[1, :] = pr[1, 0]
        buf.update_priorities(smp["step_choices"], pr.reshape(-1))
        out.update(flat(f"e{ev}/sample", dict(beta=np.float64(beta), uniforms=uni, step_choices=smp["step_choices"],
                                              weights=smp["weights"], obs=smp["obs"], rewards=smp["rewards"],
                                              priorities=pr, size=np.int64(buf.size))))
        ev += 1
    for _ in range(4):
        store()
    sample_update(0.4)
    for _ in range(5):                                              # wraps: ptr passes n_size
        store()
    sample_update(0.5); sample_update(0.7)
    store()
    sample_update(1.0)
    out["sum_tree"] = np.array([t._value for t in buf._it_sum], np.float64)
    out["min_tree"] = np.array([t._value for t in buf._it_min], np.float64)
    out["max_priority"] = np.asarray(buf._max_priority, np.float64)
    out["meta"] = np.array([n_envs, n_size, D, bs, ev])
    out["alpha"] = np.float64(alpha)
    np.savez_compressed(os.path.join(OUT, "per_buffer.npz"), **out)


def golden_pg(dist):
    """PG_Learner.update (pg_learner.py:30-71) on VanillaPolicyGradient(CategoricalActor | GaussianActor) (reinforce.py:6-31,
    pg_agent.py:36-61): a_loss = -(returns * log_prob).mean(), entropy bonus, no critic."""
    from xuance.torch.learners import PG_Learner
    from xuance.torch.rl_models.actors.categorical_actors import CategoricalActor
    from xuance.torch.rl_models.actors.gaussian_actors import GaussianActor
    from xuance.torch.rl_models.architectures.single_agent.reinforce import VanillaPolicyGradient
    torch.manual_seed(8)
    rng = np.random.default_rng(51)
    init = torch.nn.init.orthogonal_
    if dist == "categorical":
        D, A, bs = 4, 2, 96
        rep = Basic_MLP((D,), [128], None, init, nn.LeakyReLU, "cpu")
        actor = CategoricalActor(representation=rep, action_space=sp.Discrete(A), actor_hidden_size=[128], normalizer=None,
                                 initializer=init, activation=nn.LeakyReLU, device="cpu")
    else:
        D, A, bs = 17, 6, 80
        rep = Basic_Identical((D,), "cpu")
        actor = GaussianActor(representation=rep, action_space=sp.Box(-1, 1, (A,), np.float32), actor_hidden_size=[64, 64],
                              normalizer=None, initializer=init, activation=nn.ReLU, activation_action=nn.Tanh, device="cpu")
    model = VanillaPolicyGradient(actor=actor)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
    cfg = base_config(horizon_size=256, n_epochs=1, n_minibatch=1, ent_coef=0.01, end_factor_lr_decay=0.5)
    cb = Capture()
    model64 = as_double(model)
    learner = PG_Learner(cfg, model, cb)
    batches = []
    for u in range(3):
        obs = np.clip(rng.standard_normal((bs, D)), -5, 5).astype(np.float32)
        actions = rng.integers(0, A, bs).astype(np.float32) if dist == "categorical" else rng.standard_normal((bs, A)).astype(np.float32)
        batches.append(dict(obs=obs, actions=actions, returns=rng.standard_normal(bs).astype(np.float32)))
    out = run_learner_updates(learner, model, cb, batches,
                              lambda b: learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], batch_size=bs))
    out.update(float64_twin(PG_Learner(cfg, model64, Capture()), model64, batches,
                            lambda L, b: L.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], batch_size=bs)))
    out["cfg"] = np.array([cfg.learning_rate, cfg.ent_coef, cfg.grad_clip_norm, cfg.end_factor_lr_decay, learner.total_iters])
    np.savez_compressed(os.path.join(OUT, f"pg_{dist}.npz"), **out)


def golden_dueldqn_cnn():
    """DuelDQN_Learner on DuelingDeepQNetwork over Basic_CNN (dueldqn_learner.py:28-75, q_head.py:42-80, cnn.py:11-50): batch 4 of
    84x84x4 frames with 6 grey levels, three updates -> dueldqn_cnn.npz."""
    from xuance.torch.learners import DuelDQN_Learner
    from xuance.torch.rl_models.architectures.single_agent.deep_q_network import DuelingDeepQNetwork
    golden_dqn("cnn", DuelDQN_Learner, "dueldqn", DuelingDeepQNetwork, levels=6)


def golden_baseline_sizes():
    """Fixtures at the batch sizes of the BASELINE configs (C1 128, C2 8 192, C3 32 frames, C4 4 096 on 17-256-256,
    C5 32 transitions / 32 episodes x 60 steps): the split-K epilogues, multi-slab reductions and multi-tile paths of
    the HIP kernels only engage at these sizes."""
    golden_ppo("categorical", size="c1")
    golden_ppo("categorical", size="c2")
    golden_ppo("gaussian", size="c4")
    golden_dqn("cnn", size="c3")
    golden_qmix(True, size="c5")
    golden_qmix_rnn(True, size="c5")
    golden_qmix_rnn(True, fixed=True, size="c5")
    golden_ppo_chain()
    golden_ppo_cnn()
    for dist, size in (("categorical", "acrobot"), ("categorical", "lunar"), ("gaussian", "pendulum"), ("gaussian", "walker"),
                       ("categorical", "mountaincar")):
        golden_ppo(dist, size=size)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "sized":
        golden_baseline_sizes()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in globals():               # one generator by name, e.g. `golden_checkpoint decay=True`
        kw = {k: eval(v) for k, v in (a.split("=") for a in sys.argv[2:])}
        globals()[sys.argv[1]](**kw)
        sys.exit(0)
    golden_onpolicy_buffer()
    golden_offpolicy_buffer()
    golden_rms()
    golden_ppo("categorical")
    golden_ppo("gaussian")
    golden_ppo("categorical", A2C_Learner, "a2c")
    golden_ppo("gaussian", A2C_Learner, "a2c")
    golden_dqn("mlp")
    golden_dqn("cnn")
    golden_dqn("mlp", DDQN_Learner, "ddqn")
    golden_dqn("mlp", name="dqn_huber", huber=1.0)
    golden_dqn("cnn", name="dqn_huber", huber=1.0, levels=4)
    from xuance.torch.learners import DuelDQN_Learner
    from xuance.torch.rl_models.architectures.single_agent.deep_q_network import DuelingDeepQNetwork
    golden_dqn("mlp", DuelDQN_Learner, "dueldqn", DuelingDeepQNetwork)
    golden_dueldqn_cnn()
    golden_qmix(True)
    golden_qmix(False)
    golden_qmix(True, "vdn")
    golden_qmix(True, "iql")
    golden_qmix(False, "iql")
    golden_qmix_rnn(True)
    golden_qmix_rnn(False)
    golden_qmix_rnn(True, fixed=True)
    golden_qmix_rnn(True, fixed=True, rnn="LSTM")
    golden_marl_rnn_buffer()
    golden_marl_ff_buffer()
    golden_checkpoint()
    golden_checkpoint(decay=True)
    golden_per_buffer()
    golden_pg("categorical")
    golden_pg("gaussian")
    golden_ppokl("categorical")
    golden_ppokl("gaussian")
    golden_baseline_sizes()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
