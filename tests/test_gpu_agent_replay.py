"""GPU: the agent LOOPS of xuance_amd.agents.* against runs of the REFERENCE's own agents (tests/golden/agent_*.npz, recorded by
oracle/make_golden_agents.py from the unmodified PPO_Agent.train / DQN_Agent.train / QMIX_Agents.train through their callback
hooks).  The simulators' outputs come back from a tape (xuance_amd.envs.RecordedVecEnv), the reference's random decisions -- its
sampled actions, exploration coins, random actions, minibatch / replay indices -- are supplied through the agents' replay hooks;
EVERYTHING ELSE is the device loop's own work and is compared with what the reference did at the same moment: the order of
`obs_rms.update` / normalise / store (ppo_agent.py:114-128, off_policy.py:207-227), values and log-probs of the acting pass, reward
processing against `ret_rms` and the discounted-return tracker (agent.py:285-294, ppo_agent.py:144-149), path closing on
termination / truncation / buffer end (ppo_agent.py:129-160), GAE, the epsilon schedule (off_policy.py:119-127), greedy / masked
greedy actions, the update trigger (`current_step > start_training`, off_policy.py:228), ring positions, target syncs -- and the
parameters after every update phase.  Stored integers / flags / raw frames must be EQUAL, floating-point fields agree at 1e-5 of
their own scale, parameters within what gradients agreeing at 1e-5 allow (conftest.ChainCheck on the reference's own recorded
gradients)."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import load_golden, sub, assert_close, ChainCheck

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def npy(t):
    return t.detach().cpu().numpy().copy()


def categorical_uniforms(probs, acts):
    """Uniforms whose inverse-CDF image under `probs` is `acts`: the midpoint of each taken action's CDF interval (float64)."""
    p = np.asarray(probs, np.float64)
    cdf = np.cumsum(p, -1)
    a = np.asarray(acts, np.int64)[..., None]
    hi = np.take_along_axis(cdf, a, -1)[..., 0]
    lo = hi - np.take_along_axis(p, a, -1)[..., 0]
    return (0.5 * (lo + hi)).astype(np.float32)


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("kind", ["categorical", "categorical-one-launch", "categorical-one-launch-40", "gaussian", "gaussian-one-launch",
                                  "gaussian-one-launch-40", "a2c"])
def test_ppo_agent_replays_the_reference_run(kind, use_graph):
    """agent_ppo.npz: the reference's PPO_Agent (configs/ppo/classic_control/CartPole-v1.yaml) over three rollouts of 8 envs x 32
    steps with 31 terminations and 12 truncations, 2 x 2 minibatch updates per rollout.  agent_ppo_gaussian.npz: the same loop with
    configs/ppo/mujoco.yaml (Gaussian_AC on Basic_Identical: 17-256-256-{6, 1}, tanh on the mean, log_std parameter -- BASELINE
    configs[3]'s network), 25 terminations and 23 truncations, 1 x 2 updates per rollout; the sampler gets the reference's own
    normals (action - mean) / std.  use_graph: the rollout and the update phase as one captured hipGraph each (replayed on the
    following stretch of the tape / the next indices) or launch by launch."""
    from xuance_amd.agents import PPO_Agent, A2C_Agent
    from xuance_amd.envs import RecordedVecEnv, TapeCartPoleVecEnv, TapeControlVecEnv
    from xuance_amd.spaces import Box, Discrete
    # "categorical-one-launch" (round 6): the same reference run replayed through the TIMED rollout path -- xrl_rollout_cartpole_run, the
    # whole rollout as one launch of resident workgroups (csrc/rollout_actor.hip), + xrl_rollout_cartpole_values -- with the tape as the
    # kernel's provider (xrl_rollout_run_t.tape_*) and the recorded action draws as its uniforms: every assertion below is the one the
    # launches per vector step pass.  "-40": agent_ppo_40.npz, 40 envs = three actor workgroups + the bookkeeper, so the per-step
    # exchange of the observation statistics between workgroups (the tagged messages) is on the replayed path too.
    # "gaussian-one-launch[-40]": agent_ppo_gaussian[_40].npz through xrl_rollout_wide_run (csrc/rollout_wide.hip, BASELINE configs[3]'s
    # rollout kernel) with the recorded normals; its values / bootstrap values are the agent's two batched passes afterwards.
    one_launch = "-one-launch" in kind
    big = kind.endswith("-40")
    kind = kind.split("-")[0]
    gauss, a2c = kind == "gaussian", kind == "a2c"
    # a2c: agent_a2c.npz -- the reference's A2C_Agent (configs/a2c/classic_control/CartPole-v1.yaml: ActorCritic with one representation
    # per head -- nets.ActorCriticNet(head_rep_layers=1) speaks its key names --, A2C_Learner, 1 x 2 updates per rollout, no old_logp)
    g = load_golden("agent_a2c" if a2c else ("agent_ppo_gaussian_40" if big else "agent_ppo_gaussian") if gauss else "agent_ppo_40" if big else "agent_ppo")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, T, E, MB = (int(c[k]) for k in ("n_envs", "horizon_size", "n_epochs", "n_minibatch"))
    S = g["step/acts"].shape[0]
    rollouts = S // T
    A = g["step/acts"].shape[2] if gauss else 2
    if one_launch and gauss:
        env = TapeControlVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                                g["step/reset_obs"], act_dim=A, max_episode_steps=int(c["max_episode_steps"]))
    elif one_launch:
        env = TapeCartPoleVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                                 g["step/reset_obs"], max_episode_steps=int(c["max_episode_steps"]))
    else:
        env = RecordedVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                             g["step/reset_obs"], action_space=Box(-1.0, 1.0, (A,), np.float32) if gauss else Discrete(2),
                             max_episode_steps=int(c["max_episode_steps"]))
        env.prepare(T)
    net = dict(representation="Basic_Identical", representation_hidden_size=None, actor_hidden_size=[256, 256], critic_hidden_size=[256, 256],
               activation="leaky_relu", activation_action="tanh", use_fused_acting=False, use_wide_rollout=one_launch) if gauss else \
        dict(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128], critic_hidden_size=[128], activation="leaky_relu")
    cfg = Namespace(seed=1, parallels=n, running_steps=10 ** 6, horizon_size=T, n_epochs=E, n_minibatch=MB,
                    learning_rate=c["learning_rate"], vf_coef=c["vf_coef"], ent_coef=c["ent_coef"], clip_range=c["clip_range"],
                    gamma=c["gamma"], use_gae=True, gae_lambda=c["gae_lambda"], use_advnorm=True, use_grad_clip=True,
                    grad_clip_norm=c["grad_clip_norm"], use_obsnorm=True, use_rewnorm=True, obsnorm_range=c["obsnorm_range"],
                    rewnorm_range=c["rewnorm_range"], distributed_training=False, device="cuda", model_dir="/tmp/xrl_models",
                    use_hip_graph=use_graph, **net)
    agent = (A2C_Agent if a2c else PPO_Agent)(cfg, env)
    # (A2C_Learner's LinearLR runs over config.running_steps, a2c_learner.py:19-21, not over its estimate_total_iterations())
    if gauss:
        assert not agent.use_fused_rollout and (agent._wide_rollout() is not None) == one_launch
    else:
        assert agent.use_fused_rollout == one_launch and (not one_launch or (agent._actor_rollout() is not None and agent._persistent_ok()))
    assert agent.learner.total_iters == (cfg.running_steps if a2c else int(c["total_iters"]))
    init = sub(g, "init")
    assert list(getattr(agent.model, "state_keys", agent.model.ref_order)) == list(init)
    agent.model.load_state_dict(init)
    if gauss:
        noise = ((g["step/acts"] - g["step/mu"]) / g["step/std"].reshape(S, 1, A)).astype(np.float32).reshape(rollouts, T, n, A)
    else:
        noise = categorical_uniforms(g["step/probs"], g["step/acts"]).reshape(rollouts, T, n)
        assert g["step/probs"].min() > 1e-3                            # (no taken action sits in a CDF interval narrower than the tolerance)
    chain = ChainCheck(c["learning_rate"], total_iters=agent.learner.total_iters)
    f = agent.memory.soa.fields
    tm = lambda a: np.swapaxes(np.asarray(a), 0, 1)                    # the reference's env-major [n][T] -> time-major
    for p in range(rollouts):
        agent.set_action_noise(noise[p])
        agent.set_indices(g[f"phase{p}/indices"])
        agent.rollout()
        torch.cuda.synchronize()
        buf, last = sub(g, f"phase{p}/buffer"), (p + 1) * T - 1
        if gauss:     # x = mu + std z with the device's own mean (1e-5 from the reference's): not bit-equal, 1e-5 of the action scale
            assert_close(npy(f["actions"]).reshape(T, n, A), tm(buf["actions"]), 1e-5, f"rollout {p}: stored actions", scale=max(1.0, float(np.abs(buf["actions"]).max())))
        else:
            assert np.array_equal(npy(f["actions"]), tm(buf["actions"])), f"rollout {p}: stored actions"
            assert np.array_equal(npy(f["actions"]), g["step/acts"][p * T:(p + 1) * T].astype(np.float32))
        assert np.array_equal(npy(f["terminals"]) > 0, tm(buf["terminals"]) > 0), f"rollout {p}: stored terminals"
        assert_close(npy(f["observations"]).reshape(T, n, -1), tm(buf["observations"]), 1e-5, f"rollout {p}: stored (normalised) observations")
        assert_close(npy(f["rewards"]), tm(buf["rewards"]), 1e-5, f"rollout {p}: stored (processed) rewards")
        assert_close(npy(f["values"]), tm(buf["values"]), 1e-5, f"rollout {p}: stored values")
        if not a2c:
            assert_close(npy(f["aux_old_logp"]), tm(buf["old_logp"]), 1e-5, f"rollout {p}: stored old_logp")
        assert_close(npy(f["returns"]), tm(buf["returns"]), 1e-5, f"rollout {p}: returns (finish_path on termination / truncation / buffer end)")
        assert_close(npy(f["advantages"]), tm(buf["advantages"]), 1e-5, f"rollout {p}: GAE advantages", scale=float(np.abs(buf["returns"]).max()))
        # running statistics and the return tracker as the reference left them after the rollout's last vector step
        om, ov, oc = agent._obs_stats_tensors()            # (the one-launch CartPole rollout keeps its statistics in the kernel's state block)
        rm, rv, rc = (agent.pp["ret_stats"][0][0:1], agent.pp["ret_stats"][0][1:2], agent.pp["ret_count"][0]) if (one_launch and not gauss) else \
            (agent.ret_mean, agent.ret_var, agent.ret_count)
        assert_close(npy(om), g["step/obs_rms/mean"][last], 1e-5, "obs_rms.mean", scale=float(np.sqrt(g["step/obs_rms/var"][last]).max()))
        assert_close(npy(ov), g["step/obs_rms/var"][last], 1e-5, "obs_rms.var")
        assert_close(npy(oc)[0], g["step/obs_rms/count"][last], 1e-9, "obs_rms.count")
        assert_close(npy(rm)[0], g["step/ret_rms/mean"][last], 1e-5, "ret_rms.mean", scale=max(1e-3, float(np.sqrt(g["step/ret_rms/var"][last]))))
        assert_close(npy(rv)[0], g["step/ret_rms/var"][last], 1e-5, "ret_rms.var")
        assert_close(npy(rc)[0], g["step/ret_rms/count"][last], 1e-9, "ret_rms.count")
        assert_close(npy(agent.returns), g["step/returns_track"][last], 1e-5, "discounted-return tracker", scale=max(1.0, float(np.abs(g["step/returns_track"][last]).max())))
        assert agent.current_step == int(g["step/current_step"][last])
        info = agent.update()
        ref_info = sub(g, f"phase{p}/info")
        for k in (("actor-loss", "critic-loss", "entropy", "predict_value") if a2c else ("actor_loss", "critic_loss", "entropy", "predict_value")):
            assert_close(info[k], ref_info[k], 1e-5, f"phase {p} {k}",
                         scale=max(abs(float(ref_info[k])), 1.0 if k in ("actor_loss", "actor-loss", "predict_value") else 0.0))   # (means of O(1) terms of either sign)
        assert_close(info["learning_rate"], ref_info["learning_rate"], 1e-9, "learning_rate")
        assert agent.learner.iterations == int(g[f"phase{p}/iterations"])
        for u in range(E * MB):
            chain.step(sub(g, f"phase{p}/grad{u}"))
        got = {k: npy(v) for k, v in agent.model.state_dict().items()}
        chain.check(got, sub(g, f"phase{p}/param"), init, what=f"phase {p} param")
    assert (agent._rollout_graph is not None) == use_graph and (agent._update_graph is not None) == use_graph


@pytest.mark.parametrize("kind", ["dummy", "atari", "subproc"])
def test_dqn_agent_replays_the_reference_run(kind):
    """subproc (round 6): agent_dqn_subproc.npz -- the CartPole configuration below behind the reference's SubprocVecEnv (40 vector
    steps): that vector env rebinds buf_obs, so the first stored observation of the reference's train() call is what the policy acted
    on and the replay runs WITHOUT the one patched ring row the DummyVecEnv fixtures need (see `s == 0` below).
    atari: agent_dqn_atari.npz -- configs/dqn/atari.yaml (BASELINE configs[2]'s network: Basic_CNN 32/64/64 + global max-pool +
    64-512-4 on 84x84x4 uint8 frame stacks; uint8 ring), 4 envs, 22 vector steps, 9 update phases, a 12-row ring that wraps, the
    loop's Atari mode (an env that terminated without truncation keeps acting on its next observation, off_policy.py:240-242).
    agent_dqn.npz: the reference's DQN_Agent (configs/dqn/classic_control/CartPole-v1.yaml) over 64 vector steps of 8 envs: a
    16-row ring that wraps three times, 28 update phases from vector step 8 on (every second step), target syncs every 5 updates,
    epsilon from 0.5 to its floor at step 30 (frozen at the undershoot value -0.0063, off_policy.py:119-127), 41 terminations and
    7 truncations.  Per vector step: the action of every env (greedy argmax of the device's Q values, or the supplied random
    action where the supplied coin is below the device's epsilon), epsilon, current_step, the ring's write position and fill;
    per update phase: loss and parameters; at the end the whole ring, bit for bit."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import RecordedVecEnv
    from xuance_amd.spaces import Discrete
    atari, subproc = kind == "atari", kind == "subproc"
    g = load_golden("agent_dqn_atari" if atari else "agent_dqn_subproc" if subproc else "agent_dqn")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, S, B = int(c["n_envs"]), int(c["n_steps"]), int(c["batch_size"])
    A = 4 if atari else 2
    env = RecordedVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                         g["step/reset_obs"], action_space=Discrete(A), max_episode_steps=int(c["max_episode_steps"]),
                         restart=g["step/truncations"] if atari else None)
    net = dict(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64], q_hidden_size=[512]) \
        if atari else dict(representation="Basic_MLP", representation_hidden_size=[128], q_hidden_size=[128])
    cfg = Namespace(**net, activation="relu", seed=1,
                    parallels=n, running_steps=10 ** 6, buffer_size=int(c["buffer_size"]), batch_size=B, learning_rate=c["learning_rate"],
                    gamma=c["gamma"], start_greedy=c["start_greedy"], end_greedy=c["end_greedy"], decay_step_greedy=c["decay_step_greedy"],
                    sync_frequency=int(c["sync_frequency"]), training_frequency=int(c["training_frequency"]),
                    start_training=int(c["start_training"]), n_epochs=1, use_grad_clip=False, grad_clip_norm=0.5, use_obsnorm=False,
                    use_rewnorm=False, distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    agent = DQN_Agent(cfg, env)
    init = sub(g, "init")
    assert list(agent.model.ref_order) == list(init) and agent.learner.total_iters == int(c["total_iters"])
    agent.model.load_state_dict(init)
    P = int(g["n_phases"])
    agent.set_replay(coins=g["step/coin"], random_actions=g["step/random_actions"], indices=[g[f"phase{p}/indices"][0] for p in range(P)])
    mem, f = agent.memory, agent.memory.soa.fields
    trainable = [k for k in init if not k.startswith("target_")]
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))
    phase, flips = 0, 0
    for s in range(S):
        assert agent.e_greedy == g["step/eps_acted"][s] and agent.current_step == int(g["step/step_index"][s])
        slot = mem.ptr
        info = agent.train(1)
        acts = npy(env.action)
        explore = g["step/coin"][s] < np.float32(g["step/eps_acted"][s])
        assert np.array_equal(acts[explore], g["step/random_actions"][s][explore]), f"step {s}: explored actions"
        for e in np.flatnonzero(acts != g["step/acts"][s]):                        # (a greedy action may differ from the reference's only on a tie)
            q = npy(agent.model.forward(f["observations"][slot].view(n, -1), n))[e][:A]
            assert abs(q[acts[e]] - q[g["step/acts"][s][e]]) < 1e-5 * max(1.0, np.abs(q).max()), (s, e, q)
            flips += 1
        if s == 0 and subproc:       # no alias behind SubprocVecEnv: the reference stored what its policy acted on, and so did the device loop
            assert np.array_equal(g["step/obs"][0], g["raw_obs0"]) and np.array_equal(npy(f["observations"][0]).reshape(g["raw_obs0"].shape), g["raw_obs0"])
        elif s == 0:
            # The reference's first stored "obs" of a train() call is its vector env's buffer AFTER the step (an alias of
            # DummyVecEnv.buf_obs, see tests/test_oracle_agent_loops.py: test_dqn_agent_loop): the device loop stored what the
            # policy acted on; the reference's row is input data of this replay (update phases sample it until the ring wraps).
            assert np.array_equal(npy(f["observations"][0]).reshape(g["raw_obs0"].shape), g["raw_obs0"]) and np.array_equal(g["step/obs"][0], g["step/next_obs"][0])
            f["observations"][0].copy_(torch.as_tensor(g["step/obs"][0]).reshape(f["observations"][0].shape))
        assert agent.e_greedy == g["step/eps_after"][s] and agent.current_step == int(g["step/current_step"][s])
        assert mem.ptr == int(g["step/ptr"][s]) and mem.size == int(g["step/size"][s])
        if phase < P and int(g[f"phase{phase}/at_step"]) == s:
            assert agent.learner.iterations == int(g[f"phase{phase}/iterations"]), f"update trigger at step {s}"
            assert_close(info["Qloss"], g[f"phase{phase}/info/Qloss"], 1e-5, f"phase {phase} Qloss")
            assert_close(info["predictQ"], g[f"phase{phase}/info/predictQ"], 1e-5, f"phase {phase} predictQ", scale=max(1.0, abs(float(g[f"phase{phase}/info/predictQ"]))))
            chain.step({k: v for k, v in sub(g, f"phase{phase}/grad0").items()})
            ref_p = sub(g, f"phase{phase}/param")
            if ref_p:
                got = {k: npy(v) for k, v in agent.model.state_dict().items()}
                chain.check({k: got[k] for k in trainable}, {k: ref_p[k] for k in trainable}, init, what=f"phase {phase} param")
                if agent.learner.iterations % int(c["sync_frequency"]) == 0:             # hard target sync: exact copies of the eval tensors
                    for k in init:
                        if k.startswith("target_"):
                            src = k[len("target_"):] if k.startswith("target_representation") else "eval_Q_head." + k[len("target_Q_head."):]
                            assert np.array_equal(got[k], got[src]), f"phase {phase}: {k} is not the copy of {src}"
            phase += 1
        else:
            assert agent.learner.iterations == (int(g[f"phase{phase - 1}/iterations"]) if phase else 0)
    assert phase == P and flips <= 2
    fb = sub(g, "final_buffer")
    tm = lambda a: np.swapaxes(np.asarray(a), 0, 1)
    for k in ("observations", "next_observations", "actions", "rewards"):
        assert np.array_equal(npy(f[k]).reshape(tm(fb[k]).shape), tm(fb[k])), f"ring field {k}"
    assert np.array_equal(npy(f["terminals"]) > 0, tm(fb["terminals"]) > 0)


def random_action_uniforms(avail, acts):
    """Uniforms under which xrl_marl_select_actions' random choice -- the floor(u n_avail)-th available action -- is `acts`."""
    av = np.asarray(avail).reshape(-1, np.asarray(avail).shape[-1]) > 0
    a = np.asarray(acts, np.int64).reshape(-1)
    rank = np.array([int(av[r, :a[r]].sum()) for r in range(len(a))])
    return ((rank + 0.5) / np.maximum(av.sum(-1), 1)).astype(np.float32)


class _LoopTap:
    """Callback of the device loops (the learner's hooks are no-ops): collects per-step checks inside ONE train() call."""

    def __init__(self, on_step_end=None, on_epochs_end=None):
        self._se, self._ee = on_step_end, on_epochs_end

    def on_update_start(self, it, **kw):
        return {}

    def on_update_end(self, it, **kw):
        return {}

    def on_train_step_end(self, step, **kw):
        if self._se:
            self._se(step, **kw)

    def on_train_epochs_end(self, step, **kw):
        if self._ee:
            self._ee(step, **kw)


@pytest.mark.parametrize("algo", ["qmix", "vdn", "iql"])
def test_qmix_ff_agents_replay_the_reference_run(algo):
    """algo vdn / iql: agent_{vdn,iql}_ff.npz, the same run through the reference's VDN_Agents / IQL_Agents (configs/vdn|iql/sc2/3m.yaml:
    sum mixer / independent learners, no global state in the reference's buffer, IQL's own epsilon decay rule, iql_agents.py:37).
    agent_qmix_ff.npz: the reference's QMIX_Agents (configs/qmix/sc2/3m.yaml with feed-forward agents) over 36 vector steps of 4
    envs x 3 agents: one exploration coin per vector step (20 of 36 land: every agent then takes a random available action), 12
    episode ends, a 20-row ring that wraps, 16 update phases of 2 updates (double-Q, action masks, global state), target syncs every
    4 updates, epsilon 1.0 -> 0.05 with delta = (start - end) / (decay_step_greedy / n_envs) (qmix_agents.py:40).  One train(36)
    call as in the reference; per vector step (callback): actions, epsilon, ring position; per phase: losses, parameters; at the
    end the whole ring bit for bit -- including the reference's stored state after episode ends (xrl_marl_stored_state)."""
    from xuance_amd.agents import QMIX_Agents, VDN_Agents, IQL_Agents
    from xuance_amd.envs import RecordedMultiAgentVecEnv
    Agents = {"qmix": QMIX_Agents, "vdn": VDN_Agents, "iql": IQL_Agents}[algo]
    g = load_golden(f"agent_{algo}_ff")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, S, N, A, B, E = (int(c[k]) for k in ("n_envs", "n_steps", "n_agents", "n_actions", "batch_size", "n_epochs"))
    global_state, ipre = bool(g["uses_global_state"]), ("shared/" if algo == "iql" else "")
    env = RecordedMultiAgentVecEnv([dict(obs=g["acted_obs0"], state=g["acted_state0"], avail=g["acted_avail0"], at=0)],
                                   g["step/next_obs"], g["step/next_state"], g["step/next_avail"], g["step/rewards"], g["step/terminals"],
                                   g["step/truncations"], g["step/agent_mask"], g["step/reset_obs"], g["step/reset_state"],
                                   g["step/reset_avail"], g["step/episode_step"], max_episode_steps=int(c["max_episode_steps"]))
    cfg = Namespace(representation_hidden_size=[64], q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32,
                    activation="relu", seed=1, parallels=n, running_steps=10 ** 6, buffer_size=int(c["buffer_size"]), batch_size=B,
                    learning_rate=c["learning_rate"], gamma=c["gamma"], double_q=True, start_greedy=c["start_greedy"],
                    end_greedy=c["end_greedy"], decay_step_greedy=c["decay_step_greedy"], sync_frequency=int(c["sync_frequency"]),
                    training_frequency=int(c["training_frequency"]), start_training=int(c["start_training"]), n_epochs=E,
                    use_grad_clip=False, grad_clip_norm=0.5, use_actions_mask=True, use_parameter_sharing=True, use_rnn=False,
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    P = int(g["n_phases"])
    init = sub(g, "init")
    trainable = [k for k in init if not k.startswith("target_")]
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))
    st = dict(s=0, phase=0, ties=0)

    def step_end(step, **kw):
        s = st["s"]
        mem, f = agent.memory, agent.memory.soa.fields
        slot = (mem.ptr - 1) % mem.n_size
        acts = npy(f["actions"][slot]).reshape(n, N).astype(np.int64)
        if g["step/coin"][s] < np.float32(g["step/eps_acted"][s]):
            assert np.array_equal(acts, g["step/acts"][s]), f"step {s}: random available actions"
        else:
            for e, a in zip(*np.nonzero(acts != g["step/acts"][s])):             # (a greedy action may differ only on a tie)
                q = npy(agent.model.agent_plan.forward(f["obs"][slot].view(n * N, -1), agent.obs_dim, n * N))[e * N + a]
                assert abs(q[acts[e, a]] - q[g["step/acts"][s][e, a]]) < 1e-5 * max(1.0, np.abs(q).max()), (s, e, a)
                st["ties"] += 1
        if s == 0:
            # the reference's first stored obs / avail_actions of a train() call are its vector env's lists AFTER the step
            # (tests/test_oracle_agent_loops.py: test_qmix_ff_agent_loop): the device stored what the agents acted on
            assert np.array_equal(npy(f["obs"][0]).reshape(n, N, -1), g["acted_obs0"])
            assert np.array_equal(g["step/stored_obs"][0], g["step/next_obs"][0])
            f["obs"][0].copy_(torch.as_tensor(g["step/stored_obs"][0]).reshape(f["obs"][0].shape))
            f["avail_actions"][0].copy_(torch.as_tensor(g["step/stored_avail"][0]).reshape(f["avail_actions"][0].shape))
        if global_state:
            assert np.array_equal(npy(f["state"][slot]).reshape(n, -1), g["step/stored_state"][s]), f"step {s}: stored state"
        assert agent.e_greedy == g["step/eps_after"][s] and step == int(g["step/current_step"][s])
        assert mem.ptr == int(g["step/ptr"][s]) and mem.size == int(g["step/size"][s])
        st["s"] += 1

    def epochs_end(step, **kw):
        p = st["phase"]
        assert int(g[f"phase{p}/at_step"]) == st["s"], f"update trigger of phase {p}"
        assert agent.learner.iterations == int(g[f"phase{p}/iterations"])
        info = kw["update_info"]
        lk = [k for k in info if k.endswith("loss_Q")][0]                                 # (IQL's keys carry the group)
        assert_close(info[lk], g[f"phase{p}/info{E - 1}/{ipre}loss_Q"], 1e-5, f"phase {p} loss_Q")
        for e in range(E):
            chain.step(sub(g, f"phase{p}/grad{e}"))
        ref_p = sub(g, f"phase{p}/param")
        if ref_p:
            got = {k: npy(v) for k, v in agent.model.state_dict().items()}
            chain.check({k: got[k] for k in trainable}, {k: ref_p[k] for k in trainable}, init, what=f"phase {p} param")
        st["phase"] += 1

    agent = Agents(cfg, env, _LoopTap(step_end, epochs_end))
    assert [k for k in agent.model.ref_order if k in init] == list(init) and agent.learner.total_iters == int(c["total_iters"])
    agent.model.load_state_dict(init)
    explored = g["step/coin"] < g["step/eps_acted"].astype(np.float32)
    uni = np.stack([random_action_uniforms(g["step/acted_avail"][s], g["step/acts"][s]) if explored[s] else np.full(n * N, 0.5, np.float32)
                    for s in range(S)])
    agent.set_replay(coins=g["step/coin"], uniforms=uni, indices=[g[f"phase{p}/indices"][e] for p in range(P) for e in range(E)])
    agent.train(S)
    torch.cuda.synchronize()
    assert st["s"] == S and st["phase"] == P and st["ties"] <= 3
    f = agent.memory.soa.fields
    for k, v in sub(g, "final_buffer").items():
        mine = npy(f[k])                                                           # [n_size][n_envs][row]
        ref = np.swapaxes(np.asarray(v, np.float32), 0, 1).reshape(mine.shape)     # the reference: [n_envs][n_size][n_agents][...]
        assert np.array_equal(mine, ref), f"ring field {k}"


def test_qmix_rnn_agents_replay_the_reference_run():
    """agent_qmix_rnn.npz: the reference's QMIX_Agents as configs/qmix/sc2/3m.yaml ships them (Basic_RNN: fc 64 -> GRU 64 -> Q head;
    masks off, see oracle/make_golden_agents.py) through train(60) = seven run_episodes(4) calls (77 vector steps, 28 episodes, a
    16-episode ring that wraps) with six update phases of 2 updates.  Compared: every greedy action (they depend on WHICH recurrent
    rows a finished env zeroes -- the reference's own rule, QMIX_Agents.reference_rnn_reset), epsilon and current_step after every
    call (the per-finished-env update of off_policy_marl.py:532-534), ring position, losses, parameters after every phase (only
    the mixer trains: iql_learner.py:58), and the whole episode ring bit for bit, stored states included."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import RecordedMultiAgentVecEnv
    g = load_golden("agent_qmix_rnn")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, N, A, T, B, E = (int(c[k]) for k in ("n_envs", "n_agents", "n_actions", "max_episode_steps", "batch_size", "n_epochs"))
    S, P = g["step/acts"].shape[0], int(g["n_phases"])
    resets = [dict(obs=g[f"reset{i}/obs"], state=g[f"reset{i}/state"], avail=g[f"reset{i}/avail"], at=int(g[f"reset{i}/at"]))
              for i in range(int(g["n_resets"]))]
    env = RecordedMultiAgentVecEnv(resets, g["step/next_obs"], g["step/next_state"], g["step/next_avail"], g["step/rewards"],
                                   g["step/terminals"], g["step/truncations"], g["step/agent_mask"], g["step/reset_obs"],
                                   g["step/reset_state"], g["step/reset_avail"], g["step/episode_step"], max_episode_steps=T)
    cfg = Namespace(representation_hidden_size=[64], q_hidden_size=[64], fc_hidden_sizes=[64], recurrent_hidden_size=64, rnn="GRU",
                    hidden_dim_mixing_net=32, hidden_dim_hyper_net=32, activation="relu", seed=1, parallels=n, running_steps=10 ** 6,
                    buffer_size=int(c["buffer_size"]), batch_size=B, learning_rate=c["learning_rate"], gamma=c["gamma"], double_q=True,
                    start_greedy=c["start_greedy"], end_greedy=c["end_greedy"], decay_step_greedy=c["decay_step_greedy"],
                    sync_frequency=int(c["sync_frequency"]), training_frequency=1, start_training=int(c["start_training"]), n_epochs=E,
                    use_grad_clip=False, grad_clip_norm=0.5, use_actions_mask=False, use_parameter_sharing=True, use_rnn=True,
                    episode_length=T, distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    init = sub(g, "init")
    trainable = [k for k in init if not k.startswith("target_")]
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))
    st = dict(phase=0)

    def epochs_end(step, **kw):
        p = st["phase"]
        call = int(g[f"phase{p}/after_call"])
        assert step == int(g["call/current_step"][call]) and agent.e_greedy == g["call/eps"][call], f"phase {p}: loop state"
        assert agent.memory.ptr == int(g["call/ptr"][call]) and agent.memory.size == int(g["call/size"][call])
        assert agent.learner.iterations == int(g[f"phase{p}/iterations"])
        assert_close(kw["update_info"]["loss_Q"], g[f"phase{p}/info{E - 1}/loss_Q"], 1e-5, f"phase {p} loss_Q")
        for e in range(E):
            chain.step(sub(g, f"phase{p}/grad{e}"))
        got = {k: npy(v) for k, v in agent.model.state_dict().items()}
        ref_p = sub(g, f"phase{p}/param")
        chain.check({k: got[k] for k in trainable if k in chain.allow}, {k: ref_p[k] for k in trainable if k in chain.allow}, init,
                    what=f"phase {p} param")
        for k in trainable:
            if k not in chain.allow:                                               # the agent networks: no gradient in the reference
                assert np.array_equal(got[k], init[k]) and np.array_equal(ref_p[k], init[k]), k
        st["phase"] += 1

    agent = QMIX_Agents(cfg, env, _LoopTap(None, epochs_end))
    assert list(agent.model.ref_order) == list(init) and agent.learner.total_iters == int(c["total_iters"])
    agent.model.load_state_dict(init)
    explored = g["step/coin"] < g["step/eps_acted"].astype(np.float32)
    uni = np.where(explored[:, None], (g["step/acts"].reshape(S, n * N) + 0.5) / A, 0.5).astype(np.float32)   # masks off: action = floor(u A)
    agent.set_replay(coins=g["step/coin"], uniforms=uni, indices=[g[f"phase{p}/indices"][e] for p in range(P) for e in range(E)])
    # every step's actions, collected by the recorded env's own hook (the loop hands them to the simulator)
    taken = []
    step0 = env.step_device
    env.step_device = lambda: (taken.append(env.action.clone()), step0())[1]
    agent.train(int(c["train_steps"]))
    torch.cuda.synchronize()
    assert st["phase"] == P and len(taken) == S and env._n_resets == len(resets)
    acts = np.stack([npy(a) for a in taken]).astype(np.int64)
    ties = 0
    for s, e, a in zip(*np.nonzero(acts != g["step/acts"])):
        assert not explored[s], f"step {s}: a random action differs"
        ties += 1                                                                   # (a greedy action may differ only on a tie: rare)
    assert ties <= 3, ties
    assert agent.current_step == int(g["call/current_step"][-1]) and agent.e_greedy == g["call/eps"][-1]
    mem = agent.memory
    assert mem.ptr == int(g["call/ptr"][-1]) and mem.size == int(g["call/size"][-1])
    for k, v in sub(g, "final_buffer").items():
        mine = npy(mem.data[k])
        assert np.array_equal(mine.reshape(-1), np.asarray(v, np.float32).reshape(-1)), f"episode ring field {k}"


@pytest.mark.parametrize("use_graph", [True, False])
def test_pg_agent_replays_the_reference_run(use_graph):
    """agent_pg.npz: the reference's PG_Agent (configs/pg/classic_control/CartPole-v1.yaml: actor-only policy, stored values 0,
    discounted-sum returns, no advantage normalisation) over three rollouts of 8 envs x 32 steps (30 terminations, 13 truncations),
    one whole-buffer update per rollout.  What this pins beyond the PPO replay: the value that closes a CUT path -- the processed
    reward of its last step over the return statistics as they are AFTER ret_rms.update of the finished envs up to and including the
    env itself (pg_agent.py:66-79 called from on_policy.py:272-283; xrl_poststep_t.pg_bootv) -- visible in `returns`."""
    from xuance_amd.agents import PG_Agent
    from xuance_amd.envs import RecordedVecEnv
    from xuance_amd.spaces import Discrete
    g = load_golden("agent_pg")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, T = int(c["n_envs"]), int(c["horizon_size"])
    S = g["step/acts"].shape[0]
    rollouts = S // T
    env = RecordedVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                         g["step/reset_obs"], action_space=Discrete(2), max_episode_steps=int(c["max_episode_steps"]))
    env.prepare(T)
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128], activation="relu", seed=1,
                    parallels=n, running_steps=10 ** 6, horizon_size=T, n_epochs=1, n_minibatch=1, learning_rate=c["learning_rate"],
                    ent_coef=c["ent_coef"], gamma=c["gamma"], use_gae=False, gae_lambda=c["gae_lambda"], use_advnorm=False,
                    use_grad_clip=True, grad_clip_norm=c["grad_clip_norm"], use_obsnorm=True, use_rewnorm=True,
                    obsnorm_range=c["obsnorm_range"], rewnorm_range=c["rewnorm_range"], distributed_training=False, device="cuda",
                    model_dir="/tmp/xrl_models", use_hip_graph=use_graph)
    agent = PG_Agent(cfg, env)
    init = sub(g, "init")
    assert list(agent.model.ref_order) == list(init) and agent.learner.total_iters == int(c["total_iters"])
    agent.model.load_state_dict(init)
    noise = categorical_uniforms(g["step/probs"], g["step/acts"]).reshape(rollouts, T, n)
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))
    f = agent.memory.soa.fields
    tm = lambda a: np.swapaxes(np.asarray(a), 0, 1)
    for p in range(rollouts):
        agent.set_action_noise(noise[p])
        agent.set_indices(g[f"phase{p}/indices"])
        agent.rollout()
        torch.cuda.synchronize()
        buf, last = sub(g, f"phase{p}/buffer"), (p + 1) * T - 1
        assert np.array_equal(npy(f["actions"]), tm(buf["actions"])) and np.array_equal(npy(f["terminals"]) > 0, tm(buf["terminals"]) > 0)
        assert not npy(f["values"]).any() and not buf["values"].any()
        assert_close(npy(f["observations"]), tm(buf["observations"]), 1e-5, f"rollout {p}: stored observations")
        assert_close(npy(f["rewards"]), tm(buf["rewards"]), 1e-5, f"rollout {p}: stored rewards")
        assert_close(npy(f["returns"]), tm(buf["returns"]), 1e-5, f"rollout {p}: returns (cut paths closed with the processed reward)")
        assert_close(npy(f["advantages"]), tm(buf["advantages"]), 1e-5, f"rollout {p}: advantages", scale=float(np.abs(buf["returns"]).max()))
        assert_close(npy(agent.ret_var)[0], g["step/ret_rms/var"][last], 1e-5, "ret_rms.var")
        assert_close(npy(agent.returns), g["step/returns_track"][last], 1e-5, "return tracker", scale=max(1.0, float(np.abs(g["step/returns_track"][last]).max())))
        info = agent.update()
        ri = sub(g, f"phase{p}/info")
        assert_close(info["actor-loss"], ri["actor-loss"], 1e-5, f"phase {p} actor-loss", scale=max(1.0, abs(float(ri["actor-loss"]))))
        assert_close(info["entropy"], ri["entropy"], 1e-5, f"phase {p} entropy")
        chain.step(sub(g, f"phase{p}/grad0"))
        got = {k: npy(v) for k, v in agent.model.state_dict().items()}
        chain.check(got, sub(g, f"phase{p}/param"), init, what=f"phase {p} param")


def test_perdqn_agent_replays_the_reference_run():
    """agent_perdqn.npz: the reference's PerDQN_Agent (configs/perdqn/classic_control/CartPole-v1.yaml) over 64 vector steps of 8 envs,
    28 update phases on prioritized samples (2 transitions per env from each env's sum tree; the reference's `random.random()` uniforms
    are supplied), priorities <- |TD error| after every update, PER_beta after every phase, epsilon by this agent's own rule (minus delta
    per vector step, perdqn_agent.py:104-105).  Per phase: the transitions the device trees pick EQUAL the reference's, importance
    weights, loss, parameters; at the end the priority leaves and the ring."""
    from xuance_amd.agents import PerDQN_Agent
    from xuance_amd.envs import RecordedVecEnv
    from xuance_amd.spaces import Discrete
    g = load_golden("agent_perdqn")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, S, B = int(c["n_envs"]), int(c["n_steps"]), int(c["batch_size"])
    alpha, beta0 = g["per_cfg"].tolist()
    env = RecordedVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                         g["step/reset_obs"], action_space=Discrete(2), max_episode_steps=int(c["max_episode_steps"]))
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], q_hidden_size=[128], activation="relu", seed=1,
                    parallels=n, running_steps=10 ** 6, buffer_size=int(c["buffer_size"]), batch_size=B, learning_rate=c["learning_rate"],
                    gamma=c["gamma"], start_greedy=c["start_greedy"], end_greedy=c["end_greedy"], decay_step_greedy=c["decay_step_greedy"],
                    sync_frequency=int(c["sync_frequency"]), training_frequency=int(c["training_frequency"]),
                    start_training=int(c["start_training"]), n_epochs=1, use_grad_clip=False, grad_clip_norm=0.5, use_obsnorm=False,
                    use_rewnorm=False, PER_alpha=alpha, PER_beta0=beta0, distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    P = int(g["n_phases"])
    init = sub(g, "init")
    trainable = [k for k in init if not k.startswith("target_")]
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))
    st = dict(s=0, phase=0)

    def step_end(step, **kw):
        s = st["s"]
        assert np.array_equal(npy(env.action), g["step/acts"][s]), f"step {s}: actions"      # (no tie in this run)
        if s == 0:                                                                            # (the buf_obs alias of the reference's first step)
            agent.memory.soa.fields["observations"][0].copy_(torch.as_tensor(g["step/obs"][0]))
        assert agent.e_greedy == g["step/eps_after"][s] and step == int(g["step/current_step"][s])
        st["s"] += 1

    def epochs_end(step, **kw):
        p = st["phase"]
        mem = agent.memory
        assert int(g[f"phase{p}/at_step"]) == st["s"] and agent.learner.iterations == int(g[f"phase{p}/iterations"])
        assert np.array_equal(npy(mem.step_choices), g[f"phase{p}/per/step_choices"]), f"phase {p}: the transitions the trees pick"
        assert_close(npy(mem.weights), g[f"phase{p}/per/weights"], 1e-6, f"phase {p}: importance weights")   # (float32 ** in the reference under NumPy 2)
        assert agent.PER_beta == float(g[f"phase{p}/per_beta_after"])
        assert_close(kw["update_info"]["Qloss"], g[f"phase{p}/info/Qloss"], 1e-5, f"phase {p} Qloss")
        chain.step(sub(g, f"phase{p}/grad0"))
        ref_p = sub(g, f"phase{p}/param")
        if ref_p:
            got = {k: npy(v) for k, v in agent.model.state_dict().items()}
            chain.check({k: got[k] for k in trainable}, {k: ref_p[k] for k in trainable}, init, what=f"phase {p} param")
        st["phase"] += 1

    agent = PerDQN_Agent(cfg, env, _LoopTap(step_end, epochs_end))
    assert list(agent.model.ref_order) == list(init)
    agent.model.load_state_dict(init)
    agent.set_replay(coins=g["step/coin"], random_actions=g["step/random_actions"], indices=[g[f"phase{p}/per/uniforms"] for p in range(P)])
    agent.train(S)                                                # ONE call: PER_beta's increment is (1 - beta0) / train_steps
    torch.cuda.synchronize()
    assert st["s"] == S and st["phase"] == P
    mem = agent.memory
    leaves = npy(mem.it_sum)[:, mem.capacity:mem.capacity + mem.n_size]
    assert_close(leaves, g["final_priorities"], 1e-6, "priorities in the sum trees' leaves")
    assert_close(npy(mem.max_priority), g["final_max_priority"], 1e-6, "running maxima of the priorities")
    fb = sub(g, "final_buffer")
    for k in ("observations", "next_observations", "actions", "rewards"):
        assert np.array_equal(npy(mem.soa.fields[k]), np.swapaxes(fb[k], 0, 1)), f"ring field {k}"
