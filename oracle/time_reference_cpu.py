"""TEST INFRASTRUCTURE / MEASUREMENT ONLY -- the CPU baseline of record: the UNMODIFIED reference
(through oracle/ref_shim.py), device "cpu", torch threads = all cores of this container, timed

  * as whole agents: REGISTRY_Agents["PPO"] / ["QMIX"] constructed from the reference's own yaml configs and run through
    their own ``train()`` (rollout + buffer + learner updates) on host vector envs with the reference's contracts
    (xuance_amd.envs.DummyVecEnv over NumpyCartPoleEnv; DummyVecMultiAgentEnv over the SMAC-3m-shaped HostSMACLikeEnv --
    gymnasium / SMAC are not installed) -> env-steps/s, the metric of BASELINE.json;
  * per learner update at the BASELINE batch shapes.

The reference tree does not exist on the GPU box, so bench.py cannot run this there: the result is committed as
profiles/ref_cpu_baseline.json (cores stated) and bench.py reports it as `cpu_baseline` (kind "reference").
    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_cpu.py [out.json]
"""
import os, sys, time, json
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import importlib.util
spec = importlib.util.spec_from_file_location("mg", os.path.join(HERE, "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
import numpy as np
import torch
from torch import nn

torch.set_num_threads(os.cpu_count())
sp, rng = mg.sp, np.random.default_rng(0)
init = torch.nn.init.orthogonal_


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def qmix(rnn):
    from xuance.torch.rl_models.critics.base_critics import DiscreteActionValueCritic
    from xuance.torch.rl_models.representations.agent_feature import AgentFeatureEncoder
    from xuance.torch.rl_models.representations import Basic_RNN
    from xuance.torch.rl_models.modules.identity_encoder import build_identity_encoder, IdentityFeatureFusion
    N, O, S, A, B, T = 3, 30, 48, 9, 32, 60
    keys = [f"agent_{i}" for i in range(N)]
    grouping = mg.AgentGrouping.shared(keys)
    group = grouping.group_keys[0]
    if rnn:
        rep0 = Basic_RNN((O,), None, None, init, nn.ReLU, "cpu", fc_hidden_sizes=[64], recurrent_hidden_size=64,
                         N_recurrent_layers=1, dropout=0, rnn="GRU")
    else:
        rep0 = mg.Basic_MLP((O,), [64], None, init, nn.ReLU, "cpu")
    ident = build_identity_encoder(num_identities=N, mode="none", embedding_dim=None, device="cpu")
    fusion = IdentityFeatureFusion(observation_feature_dim=64, identity_feature_dim=ident.output_dim, mode="concat")
    critic = DiscreteActionValueCritic(representation=AgentFeatureEncoder(representation=rep0, identity_encoder=ident, fusion=fusion),
                                       action_space=sp.Discrete(A), critic_hidden_size=[64], normalizer=None, initializer=init,
                                       activation=nn.ReLU, device="cpu")
    model = mg.MixingQNetwork(grouping, nn.ModuleDict({group: critic}), mg.QMIX_Mixer(S, 32, 32, N, "cpu"), use_rnn=rnn, device="cpu")
    cfg = mg.base_config(learning_rate=7e-4, gamma=0.99, sync_frequency=200, start_training=0, training_frequency=1,
                         use_parameter_sharing=True, double_q=True, use_actions_mask=not rnn, use_rnn=rnn, n_epochs=8,
                         use_grad_clip=False, episode_length=T, parallels=64)
    learner = mg.QMIX_Learner(cfg, grouping, model, mg.Capture())
    if rnn:
        b = dict(obs=rng.standard_normal((B, N, T + 1, O)).astype(np.float32), actions=rng.integers(0, A, (B, N, T)).astype(np.float32),
                 rewards=rng.standard_normal((B, N, T)).astype(np.float32), terminals=np.zeros((B, N, T), bool),
                 agent_mask=np.ones((B, N, T), bool))
        sample = {k: {a: b[k][:, i] for i, a in enumerate(keys)} for k in b}
        sample.update(state=rng.standard_normal((B, T + 1, S)).astype(np.float32), filled=np.ones((B, T), bool), batch_size=B,
                      sequence_length=T)
    else:
        b = dict(obs=rng.standard_normal((B, N, O)).astype(np.float32), obs_next=rng.standard_normal((B, N, O)).astype(np.float32),
                 actions=np.zeros((B, N), np.float32), rewards=rng.standard_normal((B, N)).astype(np.float32),
                 terminals=np.zeros((B, N), bool), agent_mask=np.ones((B, N), bool), avail_actions=np.ones((B, N, A), bool),
                 avail_actions_next=np.ones((B, N, A), bool))
        sample = {k: {a: b[k][:, i] for i, a in enumerate(keys)} for k in b}
        sample.update(state=rng.standard_normal((B, S)).astype(np.float32), state_next=rng.standard_normal((B, S)).astype(np.float32),
                      batch_size=B)
    return timed(lambda: learner.update(sample), 20)


def dqn_cnn():
    rep = mg.Basic_CNN((84, 84, 4), [8, 4, 3], [4, 2, 1], [32, 64, 64], None, init, nn.ReLU, "cpu")
    model = mg.DeepQNetwork(rep, [512], sp.Discrete(4), None, init, nn.ReLU, "cpu")
    cfg = mg.base_config(learning_rate=1e-4, gamma=0.99, sync_frequency=500, start_training=0, training_frequency=1,
                         use_grad_clip=False)
    learner = mg.DQN_Learner(cfg, model, mg.Capture())
    bs = 32
    b = dict(obs=rng.integers(0, 256, (bs, 84, 84, 4)).astype(np.uint8), obs_next=rng.integers(0, 256, (bs, 84, 84, 4)).astype(np.uint8),
             actions=rng.integers(0, 4, bs).astype(np.float32), rewards=rng.standard_normal(bs).astype(np.float32),
             terminals=np.zeros(bs, np.float32))
    return timed(lambda: learner.update(batch_size=bs, **b), 10)


def ppo(bs):
    rep = mg.Basic_MLP((4,), [128], None, init, nn.LeakyReLU, "cpu")
    model = mg.SharedActorCritic(rep, mg.CategoricalActorHead(128, [128], 2, None, init, nn.LeakyReLU, "cpu"),
                                 mg.ValueHead(128, [128], None, init, nn.LeakyReLU, "cpu"))
    cfg = mg.base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, parallels=256)
    learner = mg.PPO_Learner(cfg, model, mg.Capture())
    b = dict(obs=rng.standard_normal((bs, 4)).astype(np.float32), actions=rng.integers(0, 2, bs).astype(np.float32),
             returns=rng.standard_normal(bs).astype(np.float32), advantages=rng.standard_normal(bs).astype(np.float32),
             values=rng.standard_normal(bs).astype(np.float32), aux_batch={"old_logp": np.full(bs, -0.69, np.float32)}, batch_size=bs)
    return timed(lambda: learner.update(**b), 20)


class _NullWriter:
    def __init__(self, *a, **k): pass
    def add_scalar(self, *a, **k): pass
    def add_scalars(self, *a, **k): pass


def _agent_config(yaml_rel, **over):
    import yaml
    from argparse import Namespace
    root = "/root/reference/xuance/configs"
    c = yaml.safe_load(open(os.path.join(root, "basic.yaml")))
    c.update(yaml.safe_load(open(os.path.join(root, yaml_rel))))
    c.update(device="cpu", log_dir="/tmp/xrl_ref_logs", model_dir="/tmp/xrl_ref_models", logger="tensorboard", render=False,
             render_mode="rgb_array", fps=50, test_mode=False, dl_toolbox="torch", running_steps=10 ** 8)
    c.update(over)
    return Namespace(**c)


def _median_rate(agent, steps_per_call, calls, count):
    agent.train(steps_per_call)                                   # warm-up (first rollout / first updates)
    rates = []
    for _ in range(calls):
        s0, t0 = count(), time.perf_counter()
        agent.train(steps_per_call)
        rates.append((count() - s0) / (time.perf_counter() - t0))
    return float(np.median(rates)), [round(r, 1) for r in rates]


def ppo_agent_loop(n_envs, calls=3):
    """The reference's PPO_Agent.train (ppo_agent.py:111-181) with configs/ppo/classic_control/CartPole-v1.yaml; one call =
    one rollout of 256 vector steps + 64 minibatch updates."""
    sys.path.insert(0, os.path.dirname(HERE))
    import xuance.torch.agents.base.agent as agent_mod
    from xuance.torch.agents import REGISTRY_Agents
    from xuance_amd.envs import DummyVecEnv, NumpyCartPoleEnv
    agent_mod.SummaryWriter = _NullWriter
    import tqdm as _tq
    import xuance.torch.agents.policy_gradient.ppo_agent as pa
    pa.tqdm = lambda x, *a, **k: x                                # no progress bars in the timed region
    cfg = _agent_config("ppo/classic_control/CartPole-v1.yaml", parallels=n_envs)
    envs = DummyVecEnv([NumpyCartPoleEnv] * n_envs, env_seed=1)
    envs.observation_space, envs.action_space = sp.Box(-np.inf, np.inf, (4,), np.float32), sp.Discrete(2)
    envs.reset()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs)
        rate, all_rates = _median_rate(agent, cfg.horizon_size, calls, lambda: agent.current_step)
    finally:
        os.chdir(cwd)
    return {"env_steps_per_s": round(rate, 1), "runs": all_rates, "n_envs": n_envs,
            "what": "reference PPO_Agent.train(256): 256 vector steps + 8 x 8 minibatch updates of %d" % (n_envs * 32)}


class _HostMujocoShapedEnv:
    """Host stand-in with HalfCheetah's shapes (obs 17, Box(6)): linear dynamics driven by the action plus noise, episodes of
    1 000 steps -- what tools' device provider (xuance_amd/envs/synthetic.py: SyntheticMujocoVecEnv) is on the GPU side; the
    simulator itself is third-party and not in the image, so neither side times physics."""
    max_episode_steps = 1000

    def __init__(self, env_seed=None):
        self.observation_space, self.action_space = sp.Box(-np.inf, np.inf, (17,), np.float32), sp.Box(-1.0, 1.0, (6,), np.float32)
        self.rng = np.random.default_rng(env_seed)
        self.W = (self.rng.standard_normal((6, 17)) * 0.1).astype(np.float32)
        self.state, self.steps, self.score = None, 0, 0.0

    def reset(self, seed=None):
        self.state = (self.rng.standard_normal(17) * 0.1).astype(np.float32)
        self.steps, self.score = 0, 0.0
        return self.state.copy(), {}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float32), -1, 1)
        self.state = (0.95 * self.state + a @ self.W + 0.05 * self.rng.standard_normal(17).astype(np.float32)).astype(np.float32)
        r = float(self.state[0] - 0.1 * float(a @ a))
        self.steps += 1
        self.score += r
        return self.state.copy(), r, False, self.steps >= self.max_episode_steps, {"episode_step": self.steps, "episode_score": self.score}

    def close(self):
        pass


def ppo_c4_agent_loop(n_envs=128, calls=3):
    """The reference's PPO_Agent.train with configs/ppo/mujoco.yaml (Gaussian_AC, Basic_Identical, 256-256 heads, 16 epochs x 8
    minibatches) at BASELINE configs[3]'s per-GPU size: 128 envs x horizon 256, minibatches of 4 096."""
    sys.path.insert(0, os.path.dirname(HERE))
    import xuance.torch.agents.base.agent as agent_mod
    from xuance.torch.agents import REGISTRY_Agents
    from xuance_amd.envs import DummyVecEnv
    agent_mod.SummaryWriter = _NullWriter
    import xuance.torch.agents.policy_gradient.ppo_agent as pa
    pa.tqdm = lambda x, *a, **k: x
    cfg = _agent_config("ppo/mujoco.yaml", parallels=n_envs)
    envs = DummyVecEnv([_HostMujocoShapedEnv] * n_envs, env_seed=1)
    envs.reset()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs)
        rate, all_rates = _median_rate(agent, cfg.horizon_size, calls, lambda: agent.current_step)
    finally:
        os.chdir(cwd)
    return {"env_steps_per_s": round(rate, 1), "runs": all_rates, "n_envs": n_envs,
            "what": "reference PPO_Agent.train(256) with configs/ppo/mujoco.yaml on a MuJoCo-shaped host provider: 256 vector steps of "
                    "%d envs + 16 x 8 minibatch updates of %d" % (n_envs, n_envs * 32)}


class _HostAtariShapedEnv:
    """Host stand-in with Atari's shapes (84x84x4 uint8 frame stacks, Discrete(4)): random frames, random terminations, 1 000-step
    cut-off -- what xuance_amd/envs/synthetic.py: SyntheticAtariVecEnv is on the GPU side (no emulator in the image: neither side times it)."""
    max_episode_steps = 1000

    def __init__(self, env_seed=None):
        self.observation_space, self.action_space = sp.Box(0, 255, (84, 84, 4), np.uint8), sp.Discrete(4)
        self.rng = np.random.default_rng(env_seed)
        self.steps, self.score = 0, 0.0

    def _frame(self):
        return self.rng.integers(0, 256, (84, 84, 4), dtype=np.uint8)

    def reset(self, seed=None):
        self.steps, self.score = 0, 0.0
        return self._frame(), {}

    def step(self, action):
        self.steps += 1
        r = float(self.rng.random() < 0.05)
        self.score += r
        term = bool(self.rng.random() < 0.002)
        return self._frame(), r, term, self.steps >= self.max_episode_steps, {"episode_step": self.steps, "episode_score": self.score}

    def close(self):
        pass


def dqn_c3_agent_loop(n_envs=64, steps=48, calls=3):
    """The reference's DQN_Agent.train (off_policy.py:183-262) with configs/dqn/atari.yaml at the size of BASELINE configs[2] as the
    GPU line runs it (tools/bench_secondary.py: dqn_c3): 64 envs, batch 32, ONE update per vector step (training_frequency = n_envs),
    a replay ring of 64 x 128 transitions (the yaml's 500 000 frames of host memory are not needed to time the loop)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import xuance.torch.agents.base.agent as agent_mod
    from xuance.torch.agents import REGISTRY_Agents
    from xuance_amd.envs import DummyVecEnv
    agent_mod.SummaryWriter = _NullWriter
    import xuance.torch.agents.core.off_policy as op
    op.tqdm = lambda x, *a, **k: x
    cfg = _agent_config("dqn/atari.yaml", parallels=n_envs, buffer_size=n_envs * 128, start_training=n_envs * 8,
                        training_frequency=n_envs)
    envs = DummyVecEnv([_HostAtariShapedEnv] * n_envs, env_seed=1)
    envs.reset()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs)
        rate, all_rates = _median_rate(agent, steps, calls, lambda: agent.current_step)
    finally:
        os.chdir(cwd)
    return {"env_steps_per_s": round(rate, 1), "runs": all_rates, "n_envs": n_envs,
            "what": "reference DQN_Agent.train(%d) with configs/dqn/atari.yaml on an Atari-shaped host provider: %d vector steps of %d envs, "
                    "one DQN_Learner.update (CNN, batch 32) per vector step" % (steps, steps, n_envs)}


def ppo_atari_agent_loop(n_envs=8, calls=3):
    """The reference's PPO_Agent.train with configs/ppo/atari.yaml (AC_CNN_Atari 32/64/64 + 512, Categorical_AC, uint8 rollout buffer)
    at the size tools/bench_secondary.py: ppo_atari runs: 8 envs x horizon 128, 4 epochs x 4 minibatches of 256 frames."""
    sys.path.insert(0, os.path.dirname(HERE))
    import xuance.torch.agents.base.agent as agent_mod
    from xuance.torch.agents import REGISTRY_Agents
    from xuance_amd.envs import DummyVecEnv
    agent_mod.SummaryWriter = _NullWriter
    import xuance.torch.agents.policy_gradient.ppo_agent as pa
    pa.tqdm = lambda x, *a, **k: x
    cfg = _agent_config("ppo/atari.yaml", parallels=n_envs, horizon_size=128, n_epochs=4, n_minibatch=4)
    envs = DummyVecEnv([_HostAtariShapedEnv] * n_envs, env_seed=1)
    envs.reset()
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs)
        rate, all_rates = _median_rate(agent, cfg.horizon_size, calls, lambda: agent.current_step)
    finally:
        os.chdir(cwd)
    return {"env_steps_per_s": round(rate, 1), "runs": all_rates, "n_envs": n_envs,
            "what": "reference PPO_Agent.train(128) with configs/ppo/atari.yaml on an Atari-shaped host provider: 128 vector steps of "
                    "%d envs + 4 x 4 minibatch updates of %d frames" % (n_envs, n_envs * 32)}


def qmix_agent_loop(n_envs, rnn, calls=3):
    """The reference's QMIX_Agents.train (off_policy_marl.py:310-424) with configs/qmix/sc2/3m.yaml.  Recurrent agents (the
    yaml default) with use_actions_mask as in the yaml crash in the reference's own update (iql_learner.py:78-81, see
    oracle/make_golden.py), so the recurrent run switches the mask off; the feed-forward run keeps it."""
    sys.path.insert(0, os.path.dirname(HERE))
    import xuance.torch.agents.base.agents_marl as am
    from xuance.torch.agents import REGISTRY_Agents
    from xuance_amd.envs import DummyVecMultiAgentEnv, HostSMACLikeEnv
    am.SummaryWriter = _NullWriter
    import xuance.torch.agents.core.off_policy_marl as opm
    class _Quiet:                                                  # tqdm stand-in: iterable and context manager, silent
        last_print_n = n = 0
        def __init__(self, it=None, *a, **k): self.it = it
        def __iter__(self): return iter(self.it)
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def update(self, *a, **k): pass
    opm.tqdm = _Quiet
    over = dict(parallels=n_envs, use_rnn=rnn, start_training=0 if rnn else 4 * n_envs,
                buffer_size=(5000 // n_envs) * n_envs)             # must divide by n_envs (memory_tools_marl.py:31)
    if rnn:
        over.update(use_actions_mask=False)
    else:
        over.update(representation="Basic_MLP")
    cfg = _agent_config("qmix/sc2/3m.yaml", **over)
    class _Env(HostSMACLikeEnv):
        strict_actions = not rnn                                   # (with the masks off the reference picks unavailable actions)
    envs = DummyVecMultiAgentEnv([_Env] * n_envs, env_seed=1)
    envs.observation_space = {k: sp.Box(-np.inf, np.inf, (30,), np.float32) for k in envs.agents}
    class _Disc(sp.Discrete):                                      # gymnasium's Discrete.sample (masks-off exploration, :242)
        def sample(self):
            return int(np.random.randint(self.n))
    envs.action_space = {k: _Disc(9) for k in envs.agents}
    envs.state_space = sp.Box(-np.inf, np.inf, (48,), np.float32)
    envs.groups_info = None
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        agent = REGISTRY_Agents[cfg.agent](cfg, envs)
        rate, all_rates = _median_rate(agent, 60 if rnn else 16, calls, lambda: agent.current_step)
    finally:
        os.chdir(cwd)
    return {"env_steps_per_s": round(rate, 1), "runs": all_rates, "n_envs": n_envs,
            "what": "reference QMIX_Agents.train on the SMAC-3m-shaped host env, %s, batch 32, 8 updates per %s"
                    % ("Basic_RNN/GRU, use_actions_mask off" if rnn else "Basic_MLP, action masks on",
                       "%d episodes" % n_envs if rnn else "vector step")}


if __name__ == "__main__":
    import platform
    if len(sys.argv) > 2 and sys.argv[2] == "c3":          # add the C3 loop line to an existing record (same host, same settings)
        with open(sys.argv[1]) as f:
            out = json.load(f)
        out["dqn_atari_shape_c3"] = dqn_c3_agent_loop()
        print(json.dumps(out["dqn_atari_shape_c3"]))
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "ppo_atari":   # add the PPO-Atari line to an existing record (same host, same settings)
        with open(sys.argv[1]) as f:
            out = json.load(f)
        out["ppo_atari_shape"] = ppo_atari_agent_loop()
        print(json.dumps(out["ppo_atari_shape"]))
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "c4":          # add the C4 line to an existing record (same host, same settings)
        with open(sys.argv[1]) as f:
            out = json.load(f)
        out["ppo_halfcheetah_shape_c4"] = ppo_c4_agent_loop()
        print(json.dumps(out["ppo_halfcheetah_shape_c4"]))
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)
        sys.exit(0)
    out = {"threads": os.cpu_count(), "cores": os.cpu_count(), "host": "build container (no GPU), %s" % platform.processor(),
           "torch": torch.__version__, "numpy": np.__version__,
           "ppo_cartpole": {str(n): ppo_agent_loop(n) for n in (4, 16, 256)},
           "qmix_3m_ff": qmix_agent_loop(64, False), "qmix_3m_gru": qmix_agent_loop(64, True),
           "ppo_halfcheetah_shape_c4": ppo_c4_agent_loop(), "dqn_atari_shape_c3": dqn_c3_agent_loop(),
           "ppo_atari_shape": ppo_atari_agent_loop(),
           "ppo_update_bs8192_ms": round(ppo(8192), 3),
           "qmix_ff_update_b32_ms": round(qmix(False), 3),
           "qmix_rnn_update_b32x60_ms": round(qmix(True), 3),
           "dqn_cnn_update_b32_ms": round(dqn_cnn(), 3)}
    print(json.dumps(out))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)
