"""TEST INFRASTRUCTURE ONLY -- reference-run fixtures of the AGENT LOOPS: the
UNMODIFIED reference agents (through oracle/ref_shim.py), constructed from the reference's own yaml configs by
REGISTRY_Agents[...] exactly as oracle/time_reference_cpu.py constructs them for timing, run through their own ``train()`` on
deterministic host simulators behind the reference's own vector-env classes:

  golden_agent_ppo       PPO_Agent.train           (agents/policy_gradient/ppo_agent.py:111-181, core/on_policy.py:128-205); categorical
                         (CartPole yaml) and Gaussian (mujoco yaml: BASELINE configs[3]'s network)
  golden_agent_dqn       DQN_Agent.train           (agents/core/off_policy.py:119-148, 174-270)
  golden_agent_qmix_ff   QMIX_Agents.train         (agents/core/off_policy_marl.py:112-166, 212-255, 358-424)
  golden_agent_qmix_rnn  QMIX_Agents.train -> run_episodes  (off_policy_marl.py:334-356, 426-571)

Nothing of the reference is edited.  What is recorded comes through the reference's own callback hooks (`on_train_step`,
`on_train_epochs_end`, `on_train_step_end`, callback.py:13-60) plus three instance-level wrappers that only LISTEN: the
buffer's ``sample`` (to note the indices NumPy drew -- the RNG state is saved, the call made, the state restored, the same
draws repeated and the state after the call re-installed), the agent's ``exploration`` (same trick on the torch / NumPy
streams: the coin, the uniforms, the random actions) and the learner's ``update``.  Per vector step: what the loop acted on,
the actions it took, what the simulators returned, what it stored; per update phase: the whole buffer as the learner saw it,
the indices of every minibatch, the parameters after the phase, the running statistics and epsilon.

The fixtures are replayed by tests/test_gpu_agent_replay.py through xuance_amd.agents.* on a recorded-trajectory provider
(xuance_amd/envs/recorded.py) with the recorded actions / draws / indices supplied, and by tests/test_oracle_vs_golden.py
through the oracle's restatement of the loops.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_agents.py [ppo|dqn|qmix_ff|qmix_rnn ...]
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import importlib.util

spec = importlib.util.spec_from_file_location("mg", os.path.join(HERE, "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
import numpy as np
import torch

sp, OUT = mg.sp, mg.OUT


class _NullWriter:
    def __init__(self, *a, **k): pass
    def add_scalar(self, *a, **k): pass
    def add_scalars(self, *a, **k): pass


class _Quiet:                                                      # tqdm stand-in: iterable and context manager, silent
    last_print_n = n = 0
    def __init__(self, it=None, *a, **k): self.it = it
    def __iter__(self): return iter(self.it)
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def update(self, *a, **k): pass


def agent_config(yaml_rel, **over):
    """basic.yaml + the algorithm's yaml of the reference tree + overrides (what xuance.get_arguments assembles)."""
    import yaml
    from argparse import Namespace
    root = "/root/reference/xuance/configs"
    c = yaml.safe_load(open(os.path.join(root, "basic.yaml")))
    c.update(yaml.safe_load(open(os.path.join(root, yaml_rel))))
    c.update(device="cpu", log_dir="/tmp/xrl_ref_logs", model_dir="/tmp/xrl_ref_models", logger="tensorboard", render=False,
             render_mode="rgb_array", fps=50, test_mode=False, dl_toolbox="torch", running_steps=10 ** 6)
    c.update(over)
    return Namespace(**c)


def seed_all(seed):
    import random
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def rms_np(prefix, rms):
    """RunningMeanStd (statistic_tools.py:65-110) as arrays."""
    return {f"{prefix}/mean": np.array(rms.mean, np.float64).copy(), f"{prefix}/var": np.array(rms.var, np.float64).copy(),
            f"{prefix}/count": np.float64(rms.count)}


# ------------------------------------------------------------------------------------------------------------------ PPO
class HostControlShapedEnv:
    """Host stand-in with HalfCheetah's shapes (obs 17, Box(6)): linear-tanh dynamics driven by the action plus noise, a small
    termination probability and a cut at max_episode_steps -- the simulator itself is third-party and not in the image."""
    max_episode_steps = 19

    def __init__(self, env_seed=None):
        self.observation_space, self.action_space = sp.Box(-np.inf, np.inf, (17,), np.float32), sp.Box(-1.0, 1.0, (6,), np.float32)
        self.rng = np.random.default_rng(env_seed)
        self.W = (self.rng.standard_normal((6, 17)) * 0.3).astype(np.float32)
        self.state, self.steps = None, 0

    def reset(self, seed=None):
        self.state = (self.rng.standard_normal(17) * 0.5).astype(np.float32)
        self.steps = 0
        return self.state.copy(), {}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float32), -1, 1)
        self.state = np.tanh(0.9 * self.state + a @ self.W + 0.1 * self.rng.standard_normal(17).astype(np.float32)).astype(np.float32)
        r = float(self.state[0] - 0.1 * float(a @ a))
        self.steps += 1
        term = bool(self.rng.random() < 0.04)
        return self.state.copy(), r, term, self.steps >= self.max_episode_steps, {}

    def close(self):
        pass


def golden_agent_ppo(kind="categorical"):
    """PPO_Agent over three rollouts of 8 envs x 32 steps (three buffer fills, three update phases of 2 x 2 minibatches), terminations
    AND truncations inside every rollout, behind the reference's DummyVecEnv + XuanCeEnvWrapper.
    categorical: configs/ppo/classic_control/CartPole-v1.yaml (network 4-128-{128-2, 128-1}, obs / reward normalisation on, GAE,
    advantage normalisation, clip 0.5) on CartPole simulators cut at 23 steps -> agent_ppo.npz.
    gaussian: configs/ppo/mujoco.yaml (Gaussian_AC on Basic_Identical: actor 17-256-256-6 with tanh on the mean, state-independent
    log_std, critic 17-256-256-1; BASELINE configs[3]'s network) on the HalfCheetah-shaped host env above -> agent_ppo_gaussian.npz
    (recorded next to every action: the distribution's mean and std, so that a replay can hand the sampler the very normals)."""
    from xuance.common.callback import BaseCallback
    import xuance.torch.agents.base.agent as agent_mod
    import xuance.torch.agents.policy_gradient.ppo_agent as pa
    from xuance.torch.agents import REGISTRY_Agents
    from xuance.environment.vector_envs.dummy.dummy_vec_env import DummyVecEnv
    from xuance.environment.utils.wrapper import XuanCeEnvWrapper
    from xuance_amd.envs import NumpyCartPoleEnv

    class ShortCartPole(NumpyCartPoleEnv):
        max_episode_steps = 23

    gauss, pg, a2c, big = kind in ("gaussian", "gaussian40"), kind == "pg", kind == "a2c", kind in ("categorical40", "gaussian40")
    agent_mod.SummaryWriter = _NullWriter
    pa.tqdm = lambda x, *a, **k: x
    import xuance.torch.agents.core.on_policy as onp
    onp.tqdm = lambda x, *a, **k: x
    n, T, rollouts = (40, 8, 3) if kind == "gaussian40" else (40, 16, 2) if big else (8, 32, 3)
    if kind == "gaussian40":
        # agent_ppo_gaussian_40.npz (round 6): 40 envs = three workgroups of xrl_rollout_wide_run; three rollouts of 8 steps (the host
        # env truncates at 19), one update per rollout (142 k parameters: ~4 MB)
        cfg = agent_config("ppo/mujoco.yaml", parallels=n, horizon_size=T, n_epochs=1, n_minibatch=1, seed=29)
        Env, D = HostControlShapedEnv, 17
    elif big:
        # agent_ppo_40.npz (round 6): 40 envs = three 16-env workgroups of the one-launch rollout kernel (csrc/rollout_actor.hip), so a
        # replay through that kernel also crosses its per-step exchange of observation statistics; two rollouts of 16 steps, 1 x 2 updates
        cfg = agent_config("ppo/classic_control/CartPole-v1.yaml", parallels=n, horizon_size=T, n_epochs=1, n_minibatch=2, seed=23)
        Env, D = ShortCartPole, 4
    elif pg:
        cfg = agent_config("pg/classic_control/CartPole-v1.yaml", parallels=n, horizon_size=T, seed=17)
        Env, D = ShortCartPole, 4
    elif a2c:
        cfg = agent_config("a2c/classic_control/CartPole-v1.yaml", parallels=n, horizon_size=T, n_minibatch=2, seed=19)
        Env, D = ShortCartPole, 4
    elif gauss:
        cfg = agent_config("ppo/mujoco.yaml", parallels=n, horizon_size=T, n_epochs=1, n_minibatch=2, seed=13)   # (142 k parameters: one epoch keeps the file at 5 MB)
        Env, D = HostControlShapedEnv, 17
    else:
        cfg = agent_config("ppo/classic_control/CartPole-v1.yaml", parallels=n, horizon_size=T, n_epochs=2, n_minibatch=2, seed=7)
        Env, D = ShortCartPole, 4
    seed_all(cfg.seed)
    envs = DummyVecEnv([lambda env_seed: XuanCeEnvWrapper(Env(env_seed=env_seed))] * n, 11)
    envs.observation_space = sp.Box(-np.inf, np.inf, (D,), np.float32)                    # (the shim's gymnasium types)
    envs.action_space = sp.Box(-1.0, 1.0, (6,), np.float32) if gauss else sp.Discrete(2)
    envs.reset()
    out, steps, phases = {}, [], []

    class Rec(BaseCallback):
        def on_train_step(self, current_step, **kw):
            ag = self.agent
            with torch.no_grad():
                po = ag.model(torch.as_tensor(kw["obs"]))                     # (listening only: the distribution behind the sampled actions)
                dist = dict(mu=po.distributions.mu.numpy().copy(), std=po.distributions.std.numpy().copy()) if gauss else \
                    dict(probs=po.distributions.probs.numpy().copy())
            logp = kw["aux_info"]["old_logp"] if kw["aux_info"] and "old_logp" in kw["aux_info"] else np.zeros(n)   # (PG: get_aux_info() is empty)
            steps.append(dict(obs=np.array(kw["obs"], np.float32), acts=np.array(kw["acts"]),
                              vals=np.broadcast_to(np.asarray(kw["vals"], np.float32), (n,)).copy(),       # (PG: the scalar 0, on_policy.py:160)
                              logp=np.array(logp, np.float32), next_obs=np.array(kw["next_obs"], np.float32),
                              rewards=np.array(kw["rewards"], np.float32), terminals=np.array(kw["terminals"]),
                              truncations=np.array(kw["truncations"]), **dist,
                              reset_obs=np.stack([np.asarray(i.get("reset_obs", np.zeros(D)), np.float32) for i in kw["infos"]]),
                              episode_step=np.array([i["episode_step"] for i in kw["infos"]]),
                              episode_score=np.array([i["episode_score"] for i in kw["infos"]], np.float64)))

        def on_train_epochs_end(self, current_step, **kw):
            m = kw["memory"]
            ph = dict(buffer={k: np.array(getattr(m, k)).copy() for k in ("observations", "actions", "rewards", "returns", "values",
                                                                          "terminals", "advantages")},
                      old_logp=np.array(m.auxiliary_infos["old_logp"]).copy() if m.auxiliary_infos and "old_logp" in m.auxiliary_infos else np.zeros((n, T), np.float32),
                      param=sd_np(kw["policy"] if "policy" in kw else kw["model"]),
                      info={k: np.float64(v) for k, v in kw["update_info"].items() if np.isscalar(v)},
                      indices=np.stack(self.indices), iterations=np.int64(self.agent.learner.iterations), grads=self.grads)
            self.indices, self.grads = [], []
            phases.append(ph)

        def on_train_step_end(self, current_step, **kw):
            ag = self.agent
            steps[-1].update(returns_track=np.array(ag.returns, np.float64).copy(), current_step=np.int64(current_step),
                             **rms_np("obs_rms", ag.obs_rms), **rms_np("ret_rms", ag.ret_rms))

    cb = Rec()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs, callback=cb)
    finally:
        os.chdir(cwd)
    cb.agent, cb.indices, cb.grads = agent, [], []
    sample0, update0 = agent.memory.sample, agent.learner.update
    agent.memory.sample = lambda indexes: (cb.indices.append(np.array(indexes).copy()), sample0(indexes))[1]

    def update(**samples):                                             # (listening: the clipped gradients each update stepped with)
        info = update0(**samples)
        cb.grads.append({k: p.grad.detach().numpy().copy() for k, p in agent.model.named_parameters() if p.grad is not None})
        return info
    agent.learner.update = update
    out.update(mg.flat("init", sd_np(agent.model)))
    out["raw_obs0"] = np.array(envs.buf_obs, np.float32).copy()
    agent.train(T * rollouts)
    assert len(steps) == T * rollouts and len(phases) == rollouts
    for k in steps[0]:
        out[f"step/{k}"] = np.stack([s[k] for s in steps])
    for p, ph in enumerate(phases):
        out.update(mg.flat(f"phase{p}/buffer", ph["buffer"]))
        out[f"phase{p}/buffer/old_logp"] = ph["old_logp"]
        out.update(mg.flat(f"phase{p}/param", ph["param"]))
        out.update(mg.flat(f"phase{p}/info", ph["info"]))
        out[f"phase{p}/indices"], out[f"phase{p}/iterations"] = ph["indices"], ph["iterations"]
        for u, g in enumerate(ph["grads"]):
            out.update(mg.flat(f"phase{p}/grad{u}", g))
    term, trunc = out["step/terminals"], out["step/truncations"]
    assert term.sum() > 8 and (trunc & ~term).sum() > 5, (term.sum(), trunc.sum())
    out["cfg"] = np.array([n, T, cfg.n_epochs, cfg.n_minibatch, cfg.gamma, cfg.gae_lambda, cfg.learning_rate, getattr(cfg, "vf_coef", 0.0), cfg.ent_coef,
                           getattr(cfg, "clip_range", 0.0), cfg.grad_clip_norm, cfg.obsnorm_range, cfg.rewnorm_range, agent.learner.total_iters,
                           Env.max_episode_steps], np.float64)
    out["cfg_names"] = np.array("n_envs horizon_size n_epochs n_minibatch gamma gae_lambda learning_rate vf_coef ent_coef clip_range "
                                "grad_clip_norm obsnorm_range rewnorm_range total_iters max_episode_steps".split())
    name = "agent_pg" if pg else "agent_a2c" if a2c else ("agent_ppo_gaussian_40" if big else "agent_ppo_gaussian") if gauss else "agent_ppo_40" if big else "agent_ppo"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ":", len(out), "arrays;", int(term.sum()), "terminations,", int((trunc & ~term).sum()), "truncations")


def golden_agent_ppo_gaussian():
    golden_agent_ppo("gaussian")


def golden_agent_ppo_40():
    golden_agent_ppo("categorical40")


def golden_agent_ppo_gaussian_40():
    golden_agent_ppo("gaussian40")


def golden_agent_a2c():
    """A2C_Agent (agents/policy_gradient/a2c_agent.py:18-79 on the generic loop core/on_policy.py:232-300) with
    configs/a2c/classic_control/CartPole-v1.yaml: ActorCritic with one representation per head, GAE, advantage normalisation, no
    old_logp in the buffer, one epoch x 2 minibatches per rollout -> agent_a2c.npz."""
    golden_agent_ppo("a2c")


def golden_agent_pg():
    """PG_Agent (agents/policy_gradient/pg_agent.py:12-79 on the generic loop core/on_policy.py:232-300) with
    configs/pg/classic_control/CartPole-v1.yaml: actor-only VanillaPolicyGradient, stored values 0, discounted-sum returns
    (use_gae False, advantage normalisation off), a cut path closes with the PROCESSED REWARD of its last step
    (get_terminated_values, pg_agent.py:66-79), one update per rollout -> agent_pg.npz."""
    golden_agent_ppo("pg")


# ------------------------------------------------------------------------------------------------------------------ DQN
class HostAtariShapedEnv:
    """Host stand-in with Atari's shapes (84x84x4 uint8 frame stacks, Discrete(4)): blocky frames of four grey levels (the fixture
    compresses), a small termination probability, a cut at max_episode_steps; no emulator is in the image."""
    max_episode_steps = 9

    def __init__(self, env_seed=None):
        self.observation_space, self.action_space = sp.Box(0, 255, (84, 84, 4), np.uint8), sp.Discrete(4)
        self.rng = np.random.default_rng(env_seed)
        self.steps = 0

    def _frame(self):
        return np.kron(self.rng.integers(0, 4, (7, 7, 4)) * 85, np.ones((12, 12, 1), np.int64)).astype(np.uint8)

    def reset(self, seed=None):
        self.steps = 0
        return self._frame(), {}

    def step(self, action):
        self.steps += 1
        r = float(self.rng.integers(0, 3))
        term = bool(self.rng.random() < 0.08)
        return self._frame(), r, term, self.steps >= self.max_episode_steps, {}

    def close(self):
        pass


def golden_agent_dqn_atari():
    """DQN_Agent with configs/dqn/atari.yaml (BASELINE configs[2]'s network: Basic_CNN 32/64/64 + global max-pool + 64-512-4, uint8
    replay buffer DummyOffPolicyBuffer_Atari, no normalisation) at 4 envs, a ring of 12 rows per env (it wraps), batch 8, 22 vector
    steps on the Atari-shaped host env behind DummyVecEnv_Atari.  Atari mode of the loop (off_policy.py:240-242): an env that
    TERMINATED without truncation keeps acting on its next observation -- no reset_obs, no ret_rms update, no episode count -- only a
    truncation restarts it -> agent_dqn_atari.npz."""
    golden_agent_dqn(atari=True)


def golden_agent_dqn_subproc():
    """agent_dqn.npz's run behind the reference's SubprocVecEnv instead of DummyVecEnv (40 vector steps)."""
    golden_agent_dqn(subproc=True)


def golden_agent_perdqn():
    """PerDQN_Agent (agents/qlearning_family/perdqn_agent.py:12-107) with configs/perdqn/classic_control/CartPole-v1.yaml (PER_alpha 0.5,
    PER_beta0 0.4): prioritized replay per env (memory_tools.py:471-598: batch_size / n_envs proportional draws per env from
    `random.random()`, importance weights, priorities <- |TD error| after every update), beta += (1 - beta0) / train_steps after every
    update PHASE (:72), and the agent's OWN epsilon rule -- `e_greedy -= delta` per vector step while above end_greedy (:104-105), not
    DQN_Agent's `start - current_step * delta` -- at 8 envs, a 16-row ring, batch 16 (2 per env), 64 vector steps -> agent_perdqn.npz."""
    golden_agent_dqn(per=True)


def golden_agent_dqn(atari=False, per=False, subproc=False):
    """DQN_Agent with configs/dqn/classic_control/CartPole-v1.yaml (network 4-128-128-2, MSE TD loss, no normalisation, no
    clipping) at 8 envs, a replay ring of 16 rows per env (it wraps three times), batch 16, start_training 48, an update every
    second vector step (training_frequency 16 with current_step growing by 8), hard target sync every 5 updates, epsilon from 0.5
    to 0.01 with decay_step_greedy = 1 920 (the schedule's floor is reached at vector step 30 and frozen, off_policy.py:119-127),
    64 vector steps on CartPole simulators cut at 13 steps."""
    from xuance.common.callback import BaseCallback
    import xuance.torch.agents.base.agent as agent_mod
    import xuance.torch.agents.core.off_policy as op
    from xuance.torch.agents import REGISTRY_Agents
    from xuance.environment.vector_envs.dummy.dummy_vec_env import DummyVecEnv
    from xuance.environment.utils.wrapper import XuanCeEnvWrapper
    from xuance_amd.envs import NumpyCartPoleEnv

    class ShortCartPole(NumpyCartPoleEnv):
        max_episode_steps = 13

    agent_mod.SummaryWriter = _NullWriter
    op.tqdm = lambda x, *a, **k: x
    if atari:
        from xuance.environment.vector_envs.dummy.dummy_vec_env import DummyVecEnv_Atari
        n, S, A = 4, 22, 4
        cfg = agent_config("dqn/atari.yaml", parallels=n, buffer_size=n * 12, batch_size=8, start_training=n * 3, training_frequency=2 * n,
                           sync_frequency=3, decay_step_greedy=n * n * 20, seed=9)
        seed_all(cfg.seed)
        envs = DummyVecEnv_Atari([lambda env_seed: XuanCeEnvWrapper(HostAtariShapedEnv(env_seed=env_seed))] * n, 3)
        envs.observation_space, envs.action_space = sp.Box(0, 255, (84, 84, 4), np.uint8), sp.Discrete(A)
        odt, oshape, Env = np.uint8, (84, 84, 4), HostAtariShapedEnv
    else:
        n, S, A = 8, 40 if subproc else 64, 2
        cfg = agent_config(("perdqn" if per else "dqn") + "/classic_control/CartPole-v1.yaml", parallels=n, buffer_size=n * 16, batch_size=16,
                           start_training=n * 6, training_frequency=16, sync_frequency=5,
                           decay_step_greedy=n * 30 if per else (n * n * 12 if subproc else n * n * 30), seed=21 if per else (13 if subproc else 5))
        seed_all(cfg.seed)
        if subproc:
            # the reference's OTHER vector env (environment/vector_envs/subprocess/subproc_vec_env.py: one worker process per env, pipes):
            # it rebinds buf_obs in step_wait (:117), so the `obs` the loop stores at the first step of a train() call is what the
            # policy acted on -- no alias as with DummyVecEnv (tests/test_oracle_agent_loops.py: test_dqn_agent_loop)
            from xuance.environment.vector_envs.subprocess.subproc_vec_env import SubprocVecEnv
            envs = SubprocVecEnv([lambda env_seed: XuanCeEnvWrapper(ShortCartPole(env_seed=env_seed))] * n, 3)
        else:
            envs = DummyVecEnv([lambda env_seed: XuanCeEnvWrapper(ShortCartPole(env_seed=env_seed))] * n, 3)
        envs.observation_space, envs.action_space = sp.Box(-np.inf, np.inf, (4,), np.float32), sp.Discrete(A)
        odt, oshape, Env = np.float32, (4,), ShortCartPole
    envs.reset()
    out, steps, phases = {}, [], []

    class Rec(BaseCallback):
        def on_train_step(self, current_step, **kw):
            steps.append(dict(obs=np.array(kw["obs"], odt), acts=np.array(kw["policy_out"].env_actions), next_obs=np.array(kw["next_obs"], odt),
                              rewards=np.array(kw["rewards"], np.float32), terminals=np.array(kw["terminals"]),
                              truncations=np.array(kw["truncations"]), eps_acted=np.float64(self.agent.e_greedy),
                              reset_obs=np.stack([np.asarray(i.get("reset_obs", np.zeros(oshape)), odt) for i in kw["infos"]]),
                              step_index=np.int64(current_step), **self.draw))

        def on_train_epochs_end(self, current_step, **kw):
            phases.append(dict(param=sd_np(kw["model"]), indices=np.stack(self.indices), grads=self.grads, at_step=np.int64(len(steps) - 1),
                               info={k: np.float64(v) for k, v in (self.last_info if per else kw["update_info"]).items() if np.isscalar(v) and v is not None},
                               iterations=np.int64(self.agent.learner.iterations), **({"per": self.per, "per_beta": np.float64(kw["per_beta"])} if per else {})))
            self.indices, self.grads, self.per = [], [], []

        def on_train_step_end(self, current_step, **kw):
            ag = self.agent
            steps[-1].update(eps_after=np.float64(ag.e_greedy), current_step=np.int64(current_step), ptr=np.int64(ag.memory.ptr),
                             size=np.int64(ag.memory.size))

    cb = Rec()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs, callback=cb)
    finally:
        os.chdir(cwd)
    cb.agent, cb.indices, cb.grads, cb.draw, cb.per, cb.last_info = agent, [], [], None, [], {}
    sample0, update0, explore0 = agent.memory.sample, agent.learner.update, agent.exploration

    def per_sample(beta):                                             # (listening: the uniforms behind the proportional draws, :560)
        import random
        st = random.getstate()
        smp = sample0(beta)
        after = random.getstate()
        random.setstate(st)
        k = agent.memory.batch_size // n
        uni = np.array([[random.random() for _ in range(k)] for _ in range(n)])
        random.setstate(after)
        cb.indices.append(np.stack([np.arange(n).repeat(k), smp["step_choices"].flatten()]))
        cb.per.append(dict(beta=np.float64(beta), uniforms=uni, step_choices=smp["step_choices"].copy(), weights=smp["weights"].copy()))
        return smp

    def per_update(**samples):
        td, info = update0(**samples)
        cb.grads.append({k: p.grad.detach().numpy().copy() for k, p in agent.model.named_parameters() if p.grad is not None})
        cb.per[-1]["td_error"] = np.asarray(td, np.float32).copy()
        cb.last_info = info
        return td, info

    def sample(batch_size=None):                                      # (listening: the choices NumPy made, memory_tools.py:374-377)
        st = np.random.get_state()
        smp = sample0(batch_size)
        after = np.random.get_state()
        np.random.set_state(st)
        env_c, step_c = np.random.choice(agent.memory.n_envs, agent.memory.batch_size), np.random.choice(agent.memory.size, agent.memory.batch_size)
        assert np.array_equal(smp["obs"], agent.memory.observations[env_c, step_c])
        np.random.set_state(after)
        cb.indices.append(np.stack([env_c, step_c]))
        return smp

    def update(**samples):
        info = update0(**samples)
        cb.grads.append({k: p.grad.detach().numpy().copy() for k, p in agent.model.named_parameters() if p.grad is not None})
        return info

    def exploration(pi_actions):                                      # (listening: the coin and the random actions, off_policy.py:138-141)
        st = torch.get_rng_state()
        acts = explore0(pi_actions)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        u, r = torch.rand(n), torch.randint(0, A, size=(n,))
        assert torch.equal(torch.where(u < agent.e_greedy, r, pi_actions), acts)
        torch.set_rng_state(after)
        cb.draw = dict(coin=u.numpy().copy(), random_actions=r.numpy().copy(), greedy=pi_actions.numpy().copy())
        return acts
    agent.memory.sample, agent.learner.update, agent.exploration = (per_sample if per else sample), (per_update if per else update), exploration
    out.update(mg.flat("init", sd_np(agent.model)))
    out["raw_obs0"] = np.array(envs.buf_obs, odt).copy()
    agent.train(S)
    assert len(steps) == S
    for k in steps[0]:
        out[f"step/{k}"] = np.stack([s[k] for s in steps])
    for p, ph in enumerate(phases):
        if (p % 3 == 2 or p == len(phases) - 1) if atari else (p < 2 or p % 5 == 4 or p == len(phases) - 1):   # (snapshots of some phases: every target sync is among them)
            out.update(mg.flat(f"phase{p}/param", ph["param"]))
        out.update(mg.flat(f"phase{p}/info", ph["info"]))
        out[f"phase{p}/indices"], out[f"phase{p}/iterations"], out[f"phase{p}/at_step"] = ph["indices"], ph["iterations"], ph["at_step"]
        for u, g in enumerate(ph["grads"]):
            out.update(mg.flat(f"phase{p}/grad{u}", g))
        if per:
            out.update(mg.flat(f"phase{p}/per", ph["per"][0]))
            out[f"phase{p}/per_beta_after"] = ph["per_beta"]
    out["n_phases"] = np.int64(len(phases))
    m = agent.memory
    out.update(mg.flat("final_buffer", {k: np.array(getattr(m, k)).copy() for k in ("observations", "next_observations", "actions", "rewards", "terminals")}))
    if per:                                                               # the priority trees' leaves and the running maxima
        out["final_priorities"] = np.array([[m._it_sum[i][j] for j in range(m.n_size)] for i in range(n)], np.float64)
        out["final_max_priority"] = np.array(m._max_priority, np.float64)
        out["per_cfg"] = np.array([cfg.PER_alpha, cfg.PER_beta0], np.float64)
    term, trunc = out["step/terminals"], out["step/truncations"]
    explored = out["step/coin"] < out["step/eps_acted"][:, None]
    assert term.sum() > (4 if atari else 8) and (trunc & ~term).sum() > (1 if subproc else 4) and explored.sum() > (5 if atari else 20) and \
        (~explored).sum() > (50 if atari else 120 if subproc else 200)
    assert out["step/eps_after"][-1] <= cfg.end_greedy + 1e-12 and len(np.unique(out["step/eps_after"])) > (10 if atari or subproc else 20)
    if subproc:
        assert np.array_equal(out["step/obs"][0], out["raw_obs0"]) and not np.array_equal(out["step/obs"][0], out["step/next_obs"][0])
        out["vector_env"] = np.array("SubprocVecEnv")
        envs.close()
    out["cfg"] = np.array([n, S, cfg.buffer_size, cfg.batch_size, cfg.gamma, cfg.learning_rate, cfg.start_training, cfg.training_frequency,
                           cfg.sync_frequency, cfg.start_greedy, cfg.end_greedy, cfg.decay_step_greedy, agent.learner.total_iters,
                           Env.max_episode_steps], np.float64)
    out["cfg_names"] = np.array("n_envs n_steps buffer_size batch_size gamma learning_rate start_training training_frequency sync_frequency "
                                "start_greedy end_greedy decay_step_greedy total_iters max_episode_steps".split())
    name = "agent_dqn_atari" if atari else "agent_perdqn" if per else "agent_dqn_subproc" if subproc else "agent_dqn"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ":", len(out), "arrays;", len(phases), "update phases,", int(term.sum()), "terminations,", int((trunc & ~term).sum()),
          "truncations,", int(explored.sum()), "explored actions; final epsilon", out["step/eps_after"][-1])


# ------------------------------------------------------------------------------------------------------------------ QMIX
def _smac_like_env(max_steps):
    from xuance_amd.envs import HostSMACLikeEnv

    class Env(HostSMACLikeEnv):
        max_episode_steps = max_steps
        groups_info = None

        def __init__(self, env_seed=None):
            super().__init__(env_seed)
            self.num_agents = self.n_agents
            self.env_info = {"max_episode_steps": self.max_episode_steps}
    return Env


def _stack(list_of_dicts, keys, dtype=None):
    """[n_envs] dicts keyed by agent -> [n_envs, n_agents, ...]."""
    return np.stack([[np.asarray(d[k]) for k in keys] for d in list_of_dicts]).astype(dtype) if dtype else \
        np.stack([[np.asarray(d[k]) for k in keys] for d in list_of_dicts])


def _marl_envs(Env, n, seed):
    from xuance.environment.vector_envs.dummy.dummy_vec_maenv import DummyVecMultiAgentEnv
    envs = DummyVecMultiAgentEnv([Env] * n, seed)
    envs.observation_space = {k: sp.Box(-np.inf, np.inf, (30,), np.float32) for k in envs.agents}

    class _Disc(sp.Discrete):                                          # gymnasium's Discrete.sample (masks-off exploration, :242)
        def sample(self):
            return int(np.random.randint(self.n))
    envs.action_space = {k: _Disc(9) for k in envs.agents}
    envs.state_space = sp.Box(-np.inf, np.inf, (48,), np.float32)
    return envs


def golden_agent_vdn_ff():
    golden_agent_qmix_ff("vdn")


def golden_agent_iql_ff():
    golden_agent_qmix_ff("iql")


def golden_agent_qmix_ff(algo="qmix"):
    """algo "vdn" / "iql": the same loop through VDN_Agents / IQL_Agents with configs/vdn/sc2/3m.yaml / configs/iql/sc2/3m.yaml (sum mixer /
    independent learners; no global state is stored, off_policy_marl.py:97,151; IQL's epsilon decays by (start - end) /
    decay_step_greedy per env step, iql_agents.py:37, VDN's and QMIX's by / (decay_step_greedy / n_envs), vdn_agents.py:38) ->
    agent_{vdn,iql}_ff.npz (IQL: decay_step_greedy / n_envs times shorter so that both schedules reach their floor inside the run).
    QMIX_Agents with configs/qmix/sc2/3m.yaml and representation Basic_MLP (feed-forward agents; parameter sharing, action masks,
    double-Q, global state) at 4 envs, a ring of 20 rows per env (it wraps), batch 8, start_training 16 (first update phase at
    vector step 4, `current_step >= start_training`, off_policy_marl.py:376), 2 updates every second vector step (training_frequency
    8 with current_step growing by 4), hard target sync every 4 updates, epsilon from 1.0 to 0.05 over 30 vector steps (ONE coin
    per vector step, :236), 36 vector steps on the SMAC-3m-shaped host env cut at 11 steps behind the reference's
    DummyVecMultiAgentEnv."""
    from xuance.common.callback import MultiAgentBaseCallback
    import xuance.torch.agents.base.agents_marl as am
    import xuance.torch.agents.core.off_policy_marl as opm
    from xuance.torch.agents import REGISTRY_Agents
    am.SummaryWriter = _NullWriter
    opm.tqdm = _Quiet
    n, S, N, A = 4, 36, 3, 9
    cfg = agent_config(f"{algo}/sc2/3m.yaml", parallels=n, use_rnn=False, representation="Basic_MLP", buffer_size=n * 20, batch_size=8,
                       start_training=n * 4, training_frequency=2 * n, n_epochs=2, sync_frequency=4,
                       decay_step_greedy=n * n * 30 if algo != "iql" else n * 30, seed=3)
    seed_all(cfg.seed)
    envs = _marl_envs(_smac_like_env(11), n, 21)
    envs.reset()
    keys = list(envs.agents)
    out, steps, phases = {}, [], []

    class Rec(MultiAgentBaseCallback):
        def on_train_step(self, current_step, **kw):
            info = kw["infos"]
            done = np.array([all(t.values()) or bool(tr) for t, tr in zip(kw["terminals"], kw["truncations"])])
            z_obs, z_av = {k: np.zeros(30, np.float32) for k in keys}, {k: np.zeros(A, np.float32) for k in keys}
            steps.append(dict(
                stored_obs=_stack(kw["obs"], keys, np.float32), stored_avail=_stack(kw["avail_actions"], keys, np.float32),
                stored_state=np.broadcast_to(np.asarray(kw["state"], np.float32), (n, 48)).copy() if kw["state"] is not None else np.zeros((n, 48), np.float32),
                acts=_stack(kw["acts"], keys).astype(np.int64), next_obs=_stack(kw["next_obs"], keys, np.float32),
                next_state=np.asarray(kw["next_state"] if kw["next_state"] is not None else [i["state"] for i in info], np.float32),
                next_avail=_stack(kw["next_avail_actions"], keys, np.float32),
                rewards=_stack(kw["rewards"], keys, np.float32), terminals=_stack(kw["terminals"], keys).astype(bool),
                truncations=np.asarray(kw["truncations"], bool), agent_mask=_stack([i["agent_mask"] for i in info], keys).astype(bool),
                reset_obs=_stack([i.get("reset_obs", z_obs) for i in info], keys, np.float32),
                reset_state=np.stack([np.asarray(i.get("reset_state", np.zeros(48)), np.float32) for i in info]),
                reset_avail=_stack([i.get("reset_avail_actions", z_av) for i in info], keys, np.float32),
                episode_step=np.array([i["episode_step"] for i in info], np.int64),
                done=done, eps_acted=np.float64(self.agent.e_greedy), step_index=np.int64(current_step), **self.draw))

        def on_train_epochs_end(self, current_step, **kw):
            phases.append(dict(param=sd_np(kw["model"]), indices=np.stack(self.indices), grads=self.grads, at_step=np.int64(len(steps) - 1),
                               infos=self.infos, iterations=np.int64(self.agent.learner.iterations)))
            self.indices, self.grads, self.infos = [], [], []

        def on_train_step_end(self, current_step, **kw):
            ag = self.agent
            steps[-1].update(eps_after=np.float64(ag.e_greedy), current_step=np.int64(current_step), ptr=np.int64(ag.memory.ptr),
                             size=np.int64(ag.memory.size))

    cb = Rec()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs, callback=cb)
    finally:
        os.chdir(cwd)
    cb.agent, cb.indices, cb.grads, cb.infos, cb.draw = agent, [], [], [], None
    out["acted_obs0"] = _stack(envs.buf_obs, keys, np.float32)        # what the FIRST acting pass sees (before the alias below matters)
    out["acted_avail0"] = _stack(envs.buf_avail_actions, keys, np.float32)
    out["acted_state0"] = np.asarray(envs.buf_state, np.float32).copy()
    sample0, update0, explore0 = agent.memory.sample, agent.learner.update, agent.exploration

    def sample(batch_size=None):                                      # (listening: memory_tools_marl.py:742-747)
        st = np.random.get_state()
        smp = sample0(batch_size)
        after = np.random.get_state()
        np.random.set_state(st)
        m = agent.memory
        env_c, step_c = np.random.choice(m.n_envs, m.batch_size), np.random.choice(m.size, m.batch_size)
        assert np.array_equal(smp["rewards"][keys[0]], m.data["rewards"][keys[0]][env_c, step_c])
        np.random.set_state(after)
        cb.indices.append(np.stack([env_c, step_c]))
        return smp

    def update(sample):
        info = update0(sample)
        cb.grads.append({k: p.grad.detach().numpy().copy() for k, p in agent.model.named_parameters() if p.grad is not None})
        cb.infos.append({k: np.float64(v) for k, v in info.items() if np.isscalar(v)})
        return info

    def exploration(batch_size, pi_actions_dict, avail_actions_list=None):   # (listening: ONE coin per vector step, :236; random available actions)
        st_np = np.random.get_state()
        acts = explore0(batch_size, pi_actions_dict, avail_actions_list)
        after = np.random.get_state()
        np.random.set_state(st_np)
        coin = np.random.rand()
        np.random.set_state(after)
        cb.draw = dict(coin=np.float64(coin), greedy=_stack(pi_actions_dict, keys).astype(np.int64),
                       acted_avail=_stack(avail_actions_list, keys, np.float32))
        return acts
    agent.memory.sample, agent.learner.update, agent.exploration = sample, update, exploration
    out.update(mg.flat("init", sd_np(agent.model)))
    agent.train(S)
    assert len(steps) == S
    for k in steps[0]:
        out[f"step/{k}"] = np.stack([s[k] for s in steps])
    for p, ph in enumerate(phases):
        if p < 2 or p % 4 == 3 or p == len(phases) - 1:
            out.update(mg.flat(f"phase{p}/param", ph["param"]))
        out[f"phase{p}/indices"], out[f"phase{p}/iterations"], out[f"phase{p}/at_step"] = ph["indices"], ph["iterations"], ph["at_step"]
        for u, (g, inf) in enumerate(zip(ph["grads"], ph["infos"])):
            out.update(mg.flat(f"phase{p}/grad{u}", g))
            out.update(mg.flat(f"phase{p}/info{u}", inf))
    out["n_phases"] = np.int64(len(phases))
    m = agent.memory
    for k, v in m.data.items():
        if isinstance(v, dict):
            out[f"final_buffer/{k}"] = np.stack([v[a] for a in keys], 2)          # [n_envs, n_size, n_agents, ...]
        else:
            out[f"final_buffer/{k}"] = np.array(v).copy()
    explored = out["step/coin"] < out["step/eps_acted"]
    done = out["step/done"]
    print("explored steps", int(explored.sum()), "of", S, "; episode ends", int(done.sum()), "terminated", int(out["step/terminals"].all(-1).sum()))
    assert 8 < explored.sum() < S - 8 and done.sum() > 8 and out["step/terminals"].all(-1).sum() >= 2
    out["cfg"] = np.array([n, S, N, A, cfg.buffer_size, cfg.batch_size, cfg.gamma, cfg.learning_rate, cfg.start_training, cfg.training_frequency,
                           cfg.n_epochs, cfg.sync_frequency, cfg.start_greedy, cfg.end_greedy, cfg.decay_step_greedy, agent.learner.total_iters,
                           11], np.float64)
    out["cfg_names"] = np.array("n_envs n_steps n_agents n_actions buffer_size batch_size gamma learning_rate start_training training_frequency "
                                "n_epochs sync_frequency start_greedy end_greedy decay_step_greedy total_iters max_episode_steps".split())
    out["uses_global_state"] = np.int64(bool(agent.use_global_state))
    np.savez_compressed(os.path.join(OUT, f"agent_{algo}_ff.npz"), **out)
    print(f"agent_{algo}_ff:", len(out), "arrays;", len(phases), "update phases; final epsilon", out["step/eps_after"][-1])


def golden_agent_qmix_rnn():
    """QMIX_Agents with configs/qmix/sc2/3m.yaml as shipped (Basic_RNN: fc 64 -> GRU 64, Q head 64-9; double-Q, parameter sharing,
    global state) except `use_actions_mask: False` -- with the masks on, the reference's own recurrent update raises
    (iql_learner.py:78-81 indexes a [B, T, A] mask with a [B, T+1, ...] tensor; see oracle/make_golden.py: golden_qmix_rnn) -- at 4 envs,
    episodes cut at 11 steps, a ring of 16 episodes (it wraps), batch 4 episodes, 2 updates after every run_episodes(4) call once
    current_step >= 60, target sync every 3 updates, epsilon 1.0 -> 0.05 over 200 env steps (updated per finished episode,
    off_policy_marl.py:532-534): train(60) = six run_episodes calls."""
    from xuance.common.callback import MultiAgentBaseCallback
    import xuance.torch.agents.base.agents_marl as am
    import xuance.torch.agents.core.off_policy_marl as opm
    from xuance.torch.agents import REGISTRY_Agents
    am.SummaryWriter = _NullWriter
    opm.tqdm = _Quiet
    n, N, A, L = 4, 3, 9, 11
    cfg = agent_config("qmix/sc2/3m.yaml", parallels=n, use_actions_mask=False, buffer_size=16, batch_size=4, start_training=60,
                       n_epochs=2, sync_frequency=3, decay_step_greedy=n * 200, seed=11)
    seed_all(cfg.seed)
    Env = _smac_like_env(L)
    Env.strict_actions = False                                        # (with the masks off the reference picks unavailable actions)
    envs = _marl_envs(Env, n, 31)
    keys = list(envs.agents)
    out, steps, phases, resets = {}, [], [], []

    class Rec(MultiAgentBaseCallback):
        def on_test_step(self, **kw):                                 # (run_episodes reports every step here, training mode included: :476-483)
            info = kw["infos"]
            done = np.array([all(t.values()) or bool(tr) for t, tr in zip(kw["terminals"], kw["truncations"])])
            z_obs, z_av = {k: np.zeros(30, np.float32) for k in keys}, {k: np.zeros(A, np.float32) for k in keys}
            steps.append(dict(
                acted_obs=_stack(kw["obs"], keys, np.float32), stored_state=np.broadcast_to(np.asarray(kw["state"], np.float32), (n, 48)).copy(),
                acts=_stack(kw["acts"], keys).astype(np.int64), next_obs=_stack(kw["next_obs"], keys, np.float32),
                next_state=np.asarray(kw["next_state"], np.float32), next_avail=_stack([i["avail_actions"] for i in info], keys, np.float32),
                rewards=_stack(kw["rewards"], keys, np.float32), terminals=_stack(kw["terminals"], keys).astype(bool),
                truncations=np.asarray(kw["truncations"], bool), agent_mask=_stack([i["agent_mask"] for i in info], keys).astype(bool),
                reset_obs=_stack([i.get("reset_obs", z_obs) for i in info], keys, np.float32),
                reset_state=np.stack([np.asarray(i.get("reset_state", np.zeros(48)), np.float32) for i in info]),
                reset_avail=_stack([i.get("reset_avail_actions", z_av) for i in info], keys, np.float32),
                episode_step=np.array([i["episode_step"] for i in info], np.int64), done=done, call=np.int64(len(resets) - 1),
                eps_acted=np.float64(self.eps_acted), current_step_before=np.int64(kw["current_train_step"]), **self.draw))

        def on_train_epochs_end(self, current_step, **kw):
            phases.append(dict(param=sd_np(kw["model"]), indices=np.stack(self.indices), grads=self.grads, infos=self.infos,
                               after_call=np.int64(len(resets) - 1), iterations=np.int64(self.agent.learner.iterations)))
            self.indices, self.grads, self.infos = [], [], []

    cb = Rec()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs, callback=cb)
    finally:
        os.chdir(cwd)
    cb.agent, cb.indices, cb.grads, cb.infos, cb.draw, cb.eps_acted = agent, [], [], [], None, None
    sample0, update0, explore0, reset0, run0 = agent.memory.sample, agent.learner.update, agent.exploration, envs.reset, agent.run_episodes
    calls = []

    def reset():                                                      # (listening: what every run_episodes call starts from, :436-441)
        r = reset0()
        resets.append(dict(obs=_stack(envs.buf_obs, keys, np.float32), state=np.asarray(envs.buf_state, np.float32).copy(),
                           avail=_stack(envs.buf_avail_actions, keys, np.float32), at=len(steps)))
        return r

    def run_episodes(*a, **k):
        r = run0(*a, **k)
        calls.append(dict(current_step=np.int64(agent.current_step), eps=np.float64(agent.e_greedy), ptr=np.int64(agent.memory.ptr),
                          size=np.int64(agent.memory.size), n_steps=np.int64(len(steps))))
        return r

    def sample(batch_size=None):                                      # (listening: memory_tools_marl.py:982)
        st = np.random.get_state()
        smp = sample0(batch_size)
        after = np.random.get_state()
        np.random.set_state(st)
        m = agent.memory
        ep = np.random.choice(m.size, m.batch_size)
        assert np.array_equal(smp["state"], m.data["state"][ep])
        np.random.set_state(after)
        cb.indices.append(ep)
        return smp

    def update(sample):
        info = update0(sample)
        cb.grads.append({k: p.grad.detach().numpy().copy() for k, p in agent.model.named_parameters() if p.grad is not None})
        cb.infos.append({k: np.float64(v) for k, v in info.items() if np.isscalar(v)})
        return info

    def exploration(batch_size, pi_actions_dict, avail_actions_list=None):   # (listening: the step's coin)
        st_np = np.random.get_state()
        cb.eps_acted = agent.e_greedy
        acts = explore0(batch_size, pi_actions_dict, avail_actions_list)
        after = np.random.get_state()
        np.random.set_state(st_np)
        coin = np.random.rand()
        np.random.set_state(after)
        cb.draw = dict(coin=np.float64(coin), greedy=_stack(pi_actions_dict, keys).astype(np.int64))
        return acts
    agent.memory.sample, agent.learner.update, agent.exploration, envs.reset, agent.run_episodes = sample, update, exploration, reset, run_episodes
    out.update(mg.flat("init", sd_np(agent.model)))
    agent.train(60)
    for k in steps[0]:
        out[f"step/{k}"] = np.stack([s[k] for s in steps])
    for k in calls[0]:
        out[f"call/{k}"] = np.stack([c_[k] for c_ in calls])
    for i, r in enumerate(resets):
        for k in ("obs", "state", "avail"):
            out[f"reset{i}/{k}"] = r[k]
        out[f"reset{i}/at"] = np.int64(r["at"])
    out["n_resets"], out["n_phases"] = np.int64(len(resets)), np.int64(len(phases))
    for p, ph in enumerate(phases):
        out.update(mg.flat(f"phase{p}/param", ph["param"]))
        out[f"phase{p}/indices"], out[f"phase{p}/iterations"], out[f"phase{p}/after_call"] = ph["indices"], ph["iterations"], ph["after_call"]
        for u, (g, inf) in enumerate(zip(ph["grads"], ph["infos"])):
            out.update(mg.flat(f"phase{p}/grad{u}", g))
            out.update(mg.flat(f"phase{p}/info{u}", inf))
    m = agent.memory
    for k, v in m.data.items():
        out[f"final_buffer/{k}"] = np.stack([v[a] for a in keys], 2) if isinstance(v, dict) else np.array(v).copy()   # [episodes, slots, N, ...]
    explored = out["step/coin"] < out["step/eps_acted"]
    print("calls", len(calls), "steps", len(steps), "explored", int(explored.sum()), "episodes", int(out["step/done"].sum()),
          "ring", int(m.ptr), int(m.size), "phases", len(phases), "final eps", agent.e_greedy, "current_step", agent.current_step)
    assert len(phases) >= 3 and 5 < explored.sum() < len(steps) - 5 and out["step/done"].sum() > 16
    out["cfg"] = np.array([n, N, A, L, cfg.buffer_size, cfg.batch_size, cfg.gamma, cfg.learning_rate, cfg.start_training, cfg.n_epochs,
                           cfg.sync_frequency, cfg.start_greedy, cfg.end_greedy, cfg.decay_step_greedy, agent.learner.total_iters, 60], np.float64)
    out["cfg_names"] = np.array("n_envs n_agents n_actions max_episode_steps buffer_size batch_size gamma learning_rate start_training n_epochs "
                                "sync_frequency start_greedy end_greedy decay_step_greedy total_iters train_steps".split())
    np.savez_compressed(os.path.join(OUT, "agent_qmix_rnn.npz"), **out)
    print("agent_qmix_rnn:", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(8)
    todo = sys.argv[1:] or ["ppo", "ppo_40", "ppo_gaussian", "ppo_gaussian_40", "a2c", "pg", "dqn", "dqn_subproc", "dqn_atari", "perdqn", "qmix_ff", "vdn_ff", "iql_ff", "qmix_rnn"]
    for name in todo:
        globals()[f"golden_agent_{name}"]()
