"""GPU: off-policy agent loops at the shapes of BASELINE configs C3/C5 (DQN, QMIX), the MARL replay buffer against a
NumPy mirror of memory_tools_marl.py:634-767, and the action-selection kernels on supplied randomness."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import assert_close, ChainCheck

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_egreedy_and_marl_select_on_supplied_randomness(oracle):
    from xuance_amd import ops
    rng = np.random.default_rng(0)
    n, A = 300, 5
    q = rng.standard_normal((n, A)).astype(np.float32)
    u = rng.random(n).astype(np.float32)
    ra = rng.integers(0, A, n).astype(np.int32)
    eps = torch.tensor([0.3], device="cuda")
    act = torch.zeros(n, dtype=torch.int32, device="cuda")
    act_f = torch.zeros(n, device="cuda")
    ops.egreedy(q=torch.as_tensor(q).cuda(), uniforms=torch.as_tensor(u).cuda(), randoms=torch.as_tensor(ra).cuda(),
                eps_dev=eps, action=act, action_f=act_f, n=n, A=A, ld=A, seed=1, step=0, step_dev=None)
    exp = oracle.egreedy_select(q.argmax(1), ra, u, np.float32(0.3))         # off_policy.py:138-141
    assert np.array_equal(act.cpu().numpy(), exp) and np.array_equal(act_f.cpu().numpy(), exp.astype(np.float32))
    # MARL: masked greedy when the coin does not land, uniformly random AVAILABLE action when it does
    R = 240
    qm = rng.standard_normal((R, 9)).astype(np.float32)
    av = (rng.random((R, 9)) < 0.6).astype(np.float32); av[:, 0] = 1
    um = rng.random(R).astype(np.float32)
    out = torch.zeros(R, dtype=torch.int32, device="cuda")
    for coin, explore in ((0.9, False), (0.1, True)):
        ops.marl_select_actions(q=torch.as_tensor(qm).cuda(), avail=torch.as_tensor(av).cuda(), eps_dev=eps,
                                coin=torch.tensor([coin], device="cuda"), uniforms=torch.as_tensor(um).cuda(), action=out,
                                action_f=None, R=R, A=9, ld=9, seed=1, step=0, step_dev=None)
        got = out.cpu().numpy()
        if not explore:
            assert np.array_equal(got, np.where(av > 0, qm, -1e10).argmax(1))   # value_factorization.py:87-92
        else:
            for r in range(R):
                idxs = np.flatnonzero(av[r])
                assert got[r] == idxs[min(int(np.float32(um[r]) * np.float32(len(idxs))), len(idxs) - 1)]


def test_marl_buffer_matches_numpy_mirror():
    from xuance_amd.memory_marl import HipMARLOffPolicyBuffer
    from xuance_amd.spaces import Box, Discrete
    rng = np.random.default_rng(1)
    keys = ["agent_0", "agent_1", "agent_2"]
    n_envs, n_size, O, S, A, steps, bs = 4, 6, 5, 7, 4, 9, 10
    buf = HipMARLOffPolicyBuffer(keys, Box(-1, 1, (S,)), {k: Box(-1, 1, (O,)) for k in keys}, {k: Discrete(A) for k in keys},
                                 n_envs, n_envs * n_size, bs, use_actions_mask=True,
                                 avail_actions_shape={k: (A,) for k in keys})
    ref = {}

    def ref_store(t, data):                                    # memory_tools_marl.py:731-740 (env-major arrays)
        p = t % n_size
        for k, v in data.items():
            if k in ("state", "state_next"):
                ref.setdefault(k, np.zeros((n_envs, n_size) + v.shape[1:], np.float32))[:, p] = v
            else:
                for a in keys:
                    ref.setdefault((k, a), np.zeros((n_envs, n_size) + v[a].shape[1:], np.float32))[:, p] = v[a]
    for t in range(steps):
        data = {"obs": {a: rng.standard_normal((n_envs, O)).astype(np.float32) for a in keys},
                "obs_next": {a: rng.standard_normal((n_envs, O)).astype(np.float32) for a in keys},
                "actions": {a: rng.integers(0, A, n_envs).astype(np.float32) for a in keys},
                "rewards": {a: rng.standard_normal(n_envs).astype(np.float32) for a in keys},
                "terminals": {a: rng.random(n_envs) < 0.2 for a in keys},
                "agent_mask": {a: rng.random(n_envs) < 0.9 for a in keys},
                "avail_actions": {a: rng.random((n_envs, A)) < 0.7 for a in keys},
                "avail_actions_next": {a: rng.random((n_envs, A)) < 0.7 for a in keys},
                "state": rng.standard_normal((n_envs, S)).astype(np.float32),
                "state_next": rng.standard_normal((n_envs, S)).astype(np.float32)}
        buf.store(**data)
        ref_store(t, data)
    assert (buf.ptr, buf.size) == (steps % n_size, n_size)
    np.random.seed(5)
    env = np.random.choice(n_envs, bs); step = np.random.choice(n_size, bs)
    np.random.seed(5)
    s = buf.sample()
    for k in ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions", "avail_actions_next"):
        for a in keys:
            assert np.array_equal(s[k][a].cpu().numpy(), ref[(k, a)][env, step].astype(np.float32)), (k, a)
    assert np.array_equal(s["state"].cpu().numpy(), ref["state"][env, step])
    assert s["batch_size"] == bs


def test_marl_buffer_vs_reference_fixture():
    """HipMARLOffPolicyBuffer against the unmodified MARL_OffPolicyBuffer (tests/golden/marl_ff_buffer.npz): ring contents
    after a wrapping sequence of stores (reference nested agent dicts in, bool fields as 0/1) and sample() on the same
    NumPy global-RNG draws."""
    from conftest import load_golden, sub
    from xuance_amd.memory_marl import HipMARLOffPolicyBuffer
    from xuance_amd.spaces import Box, Discrete
    g = load_golden("marl_ff_buffer")
    n_envs, n_size, N, O, A, S, bs, n_steps = (int(x) for x in g["meta"])
    keys = [f"agent_{i}" for i in range(N)]
    buf = HipMARLOffPolicyBuffer(keys, Box(-1, 1, (S,)), {k: Box(-1, 1, (O,)) for k in keys}, {k: Discrete(A) for k in keys},
                                 n_envs, n_envs * n_size, bs, use_actions_mask=True,
                                 avail_actions_shape={k: (A,) for k in keys})
    per_agent = ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions", "avail_actions_next")
    for t in range(n_steps):
        d = sub(g, f"t{t}")
        step = {k: {a: d[k][:, i] for i, a in enumerate(keys)} for k in per_agent}
        buf.store(state=d["state"], state_next=d["state_next"], **step)
        assert [buf.ptr, buf.size] == d["ptr_size"].tolist()
    np.random.seed(11)
    s = buf.sample()
    for k in per_agent:
        ref = g[f"sample/{k}"].astype(np.float32)
        for i, a in enumerate(keys):
            assert np.array_equal(s[k][a].cpu().numpy(), ref[:, i]), (k, a)
    for k in ("state", "state_next"):
        assert np.array_equal(s[k].cpu().numpy(), g[f"sample/{k}"]), k
    assert s["batch_size"] == bs
    # the whole ring, through gathers of every (env, step) cell
    idx = torch.arange(n_envs * n_size, device="cuda")
    full = buf.sample(indexes=idx)                             # flat index = env * n_size + step (memory_marl.py)
    for k in per_agent:
        ref = g[f"data/{k}"].astype(np.float32).reshape((n_envs * n_size,) + g[f"data/{k}"].shape[2:])
        for i, a in enumerate(keys):
            assert np.array_equal(full[k][a].cpu().numpy(), ref[:, i]), (k, a)


def test_dqn_agent_learns_cartpole():
    """C3-style loop (store -> sample -> TD update with target sync, epsilon decay) on the device CartPole."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], q_hidden_size=[128], activation="relu",
                    seed=1, parallels=16, running_steps=200000, buffer_size=16 * 2000, batch_size=256, learning_rate=1e-3,
                    gamma=0.99, start_greedy=0.5, end_greedy=0.01, decay_step_greedy=20000, sync_frequency=50,
                    training_frequency=16, start_training=1000, use_grad_clip=False, grad_clip_norm=0.5,
                    use_obsnorm=False, use_rewnorm=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
    env = DeviceCartPoleVecEnv(16, seed=1)
    agent = DQN_Agent(cfg, env)
    agent.train(300)
    e0, s0, _ = env.episode_stats()
    env.stats.zero_()
    info = agent.train(2500)
    e1, s1, _ = env.episode_stats()
    assert np.isfinite(info["Qloss"]) and agent.learner.iterations > 1000
    assert abs(agent.e_greedy - 0.01) < 0.02                  # reaches end_greedy after decay_step/n_envs vector steps
    assert s1 > 2 * s0 and s1 > 50, (s0, s1)


def test_qmix_agents_on_smac_3m_shape():
    """C5 shapes: 3 agents, obs 30, state 48, 9 masked actions, 64 envs, batch 32, n_epochs 8 (configs/qmix/sc2/3m.yaml)."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = Namespace(representation_hidden_size=[64], q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32,
                    activation="relu", seed=1, parallels=64, running_steps=10 ** 6, buffer_size=64 * 64, batch_size=32,
                    learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05,
                    decay_step_greedy=50000, sync_frequency=200, training_frequency=1, start_training=640, n_epochs=8,
                    use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True,
                    use_rnn=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
    env = SyntheticSMACVecEnv(64, seed=3)
    agent = QMIX_Agents(cfg, env)
    assert agent.model.params.P >= 17258                      # trainable parameters of the FF 3m model
    p0 = agent.model.params.flat.clone()
    info = agent.train(40)
    assert agent.learner.iterations == 8 * 30                 # updates start at current_step >= 640 (vector step 10)
    assert np.isfinite(info["loss_Q"]) and np.isfinite(info["predictQ"])
    assert float((agent.model.params.flat - p0).abs().max()) > 0
    # every stored action was available when it was taken
    f = agent.memory.soa.fields
    acts = f["actions"][:40].long().view(40, 64, 3)
    av = f["avail_actions"][:40].view(40, 64, 3, 9)
    assert bool(av.gather(-1, acts.unsqueeze(-1)).all())


def test_qmix_ff_loop_at_c5_size_vs_oracle(oracle):
    """The feed-forward QMIX LOOP at the C5 size (64 envs x 3 agents, batch 32, 8 updates per vector step as one graph launch:
    what tools/bench_secondary.py times) replayed by the oracle: every acting step's actions = the masked argmax of the
    oracle's own Q values under the parameters of that moment (epsilon = 0; ties within 1e-5 may fall either way), and the three
    update phases -- 24 chained updates with two hard target syncs, batches drawn by the device's Philox stream from the
    filling ring (the indices are the fixed input) -- against the oracle's chain: losses per phase, parameters at the end within
    what gradients agreeing at 1e-5 of their scale allow (conftest.ChainCheck)."""
    from xuance_amd import ops
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    np.random.seed(0)
    n, N, A, K, E = 64, 3, 9, 12, 8
    cfg = Namespace(representation_hidden_size=[64], q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32,
                    activation="relu", seed=1, parallels=n, running_steps=10 ** 6, buffer_size=n * 64, batch_size=32,
                    learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=0.0, end_greedy=0.0,
                    decay_step_greedy=50000, sync_frequency=10, training_frequency=1, start_training=n * K, n_epochs=E,
                    use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True,
                    use_rnn=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
    agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(n, seed=3))
    mem, lr, net = agent.memory, agent.learner, agent.model
    assert lr.fused_eligible() and agent.use_graph_updates
    sd = {k: v.cpu().numpy().copy() for k, v in net.state_dict().items()}
    sd0 = {k: v.copy() for k, v in sd.items()}
    agent.train(K)
    assert lr.iterations == 0
    infos = [agent.train(1) for _ in range(3)]                        # three vector steps, each followed by an 8-update phase
    torch.cuda.synchronize()
    assert lr.iterations == 3 * E and mem.size == K + 3
    f = {k: v.cpu().numpy() for k, v in mem.soa.fields.items()}        # ring [n_size][n_envs][row]
    opt = oracle.AdamOracle({k: sd[k] for k in net.ref_order if not k.startswith("target_")}, lr=7e-4, eps=1e-5, total_iters=lr.total_iters)
    chain = ChainCheck(7e-4, total_iters=lr.total_iters)
    ocfg = dict(gamma=0.99, double_q=True, use_actions_mask=True)
    keys = agent.agent_keys
    idx = torch.zeros(32, dtype=torch.int64, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    size_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    pe = f"individual_q_networks.{net.group}"
    ties = 0
    for s in range(K + 3):
        # acting of vector step s under the oracle's parameters of that moment
        obs = f["obs"][s].reshape(n * N, -1)
        h = oracle.MLP(oracle.collect_seq(sd, f"{pe}.representation.obs_representation.model", "relu", last_act="relu")).forward(obs)
        q = oracle.MLP(oracle.collect_seq(sd, f"{pe}.critic_head.q_value", "relu")).forward(h)
        av = f["avail_actions"][s].reshape(n * N, A)
        greedy = np.where(av > 0, q, -1e10).argmax(-1)                 # value_factorization.py:87-90
        acted = f["actions"][s].reshape(n * N).astype(np.int64)
        assert (av[np.arange(n * N), acted] > 0).all()
        for r in np.flatnonzero(greedy != acted):
            assert abs(q[r, greedy[r]] - q[r, acted[r]]) < 1e-5 * max(1.0, np.abs(q[r]).max()), f"step {s} row {r}: not the masked argmax"
            ties += 1
        if s < K:
            continue
        p = s - K
        size_t.fill_(s + 1)                                            # the ring held s + 1 rows when this phase drew
        for e in range(E):
            ops.sample_replay_indices(idx, mem.n_envs, mem.n_size, size_t, agent.seed, 0, ctr)
            ops.counter_add(ctr, 1)
            assert int((idx % mem.n_size).max()) <= s
            smp = mem.sample(indexes=idx.clone())
            st = lambda k: np.stack([smp[k][a].cpu().numpy() for a in keys], 1)
            b = {k: st(k) for k in ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions",
                                    "avail_actions_next")}
            b.update(state=smp["state"].cpu().numpy(), state_next=smp["state_next"].cpu().numpy())
            oi, grads = oracle.qmix_forward_backward(sd, b, ocfg, group=net.group)
            chain.step(grads)
            opt.step(grads)
            if (p * E + e + 1) % 10 == 0:
                oracle.qmix_copy_target(sd)
        assert_close(infos[p]["loss_Q"], oi["loss"], 1e-5, f"loss_Q of phase {p}")
        assert_close(infos[p]["predictQ"], oi["predictQ"], 1e-5, f"predictQ of phase {p}", scale=float(np.abs(oi["q_tot_eval"]).mean()))
    assert ties <= 4, ties
    got = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
    chain.check({k: got[k] for k in chain.allow}, {k: sd[k] for k in chain.allow}, sd0)
    # target copies = the eval tensors as they were at update 20: same propagated bound (of the whole chain: slightly generous)
    ev = lambda k: k[len("target_"):] if k.startswith("target_individual") else "eval_Qtot." + k[len("target_Qtot."):]
    tk = [k for k in sd if k.startswith("target_") and ev(k) in chain.allow]
    assert tk
    chain.check({ev(k): got[k] for k in tk}, {ev(k): sd[k] for k in tk}, sd0, what="target copy of")


def test_dqn_agent_on_atari_shape():
    """C3 shapes: 64 envs of 84x84x4 uint8 frames, SoA uint8 replay in HBM (2 x 28 224 B per transition), batch 32."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = Namespace(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                    q_hidden_size=[512], activation="relu", seed=1, parallels=64, running_steps=10 ** 6,
                    buffer_size=64 * 32, batch_size=32, learning_rate=1e-4, gamma=0.99, start_greedy=0.5, end_greedy=0.05,
                    decay_step_greedy=10 ** 6, sync_frequency=500, training_frequency=64, start_training=64 * 8,
                    use_grad_clip=False, grad_clip_norm=0.5, use_obsnorm=False, use_rewnorm=False,
                    distributed_training=False, device="cuda", model_dir="/tmp/x")
    env = SyntheticAtariVecEnv(64, seed=2)
    agent = DQN_Agent(cfg, env)
    assert agent.memory.soa.fields["observations"].dtype == torch.uint8
    assert agent.memory.soa.row_bytes["observations"] == 28224
    p0 = agent.model.params.flat.clone()
    info = agent.train(14)
    assert agent.learner.iterations == 5 and np.isfinite(info["Qloss"])
    assert float((agent.model.params.flat - p0).abs().max()) > 0
    # the frames written to the ring are the frames the env produced (last stored step)
    t = (agent.memory.ptr - 1) % agent.memory.n_size
    assert torch.equal(agent.memory.soa.fields["next_observations"][t], env.next_obs)


def test_dqn_vector_step_pair_graph_equals_the_launch_by_launch_loop():
    """DQN_Agent.train with two vector steps per graph launch (acting with epsilon from the step counter, provider, ring store with
    the slot from the counter, update phase -- DQN_Agent._run_pair) against the launch-by-launch loop: identical parameters, target,
    ring contents, counters and epsilon after the same number of steps -- across the ring's wrap-around, the end of the epsilon decay
    and an odd number of steps (the launch-by-launch loop takes over between pairs)."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n = 16
    out = []
    for pair in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = Namespace(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                        q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 6,
                        buffer_size=n * 24, batch_size=32, learning_rate=1e-4, gamma=0.99, start_greedy=0.5, end_greedy=0.05,
                        decay_step_greedy=n * n * 30, sync_frequency=7, training_frequency=1, start_training=n * 4,
                        use_grad_clip=True, grad_clip_norm=0.5, use_obsnorm=False, use_rewnorm=False,
                        distributed_training=False, device="cuda", model_dir="/tmp/x", use_step_graph=pair)
        env = SyntheticAtariVecEnv(n, seed=2, max_episode_steps=11)
        agent = DQN_Agent(cfg, env)
        infos = [agent.train(9), agent.train(21), agent.train(12)]         # 42 steps: ring of 24 slots wraps, epsilon bottoms out at ~30
        torch.cuda.synchronize()
        assert (getattr(agent, "_pair_graph", None) is not None) == pair
        mem = agent.memory
        out.append(dict(params=agent.model.params.flat.cpu().numpy().copy(), target=agent.model.target_flat.cpu().numpy().copy(),
                        fields={k: v.cpu().numpy().copy() for k, v in mem.soa.fields.items()}, ptr=mem.ptr, size=mem.size,
                        size_dev=int(mem.size_dev.item()), eps=agent.e_greedy, cs=agent.current_step, hs=agent._host_step,
                        it=agent.learner.iterations, obs=env.buf_obs.cpu().numpy().copy(), steps=env.steps.cpu().numpy().copy(),
                        q=[(i["Qloss"], i["predictQ"]) for i in infos], opt=int(agent.learner.optimizer.read().step)))
    a, b = out
    for k in ("ptr", "size", "size_dev", "eps", "cs", "hs", "it", "q", "opt"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["eps"] <= 0.05 and a["size"] == 24
    for k in ("params", "target", "obs", "steps"):
        assert np.array_equal(a[k], b[k]), k
    for k in a["fields"]:
        assert np.array_equal(a["fields"][k], b["fields"][k]), k


@pytest.mark.parametrize("mode", ["ring", "gathered", "layered"])
def test_qmix_graph_update_phase_equals_eager_updates_on_the_same_indices(mode):
    """QMIX_Learner.update_from_buffer (device sampling + gather + update, n_epochs per hipGraph launch) vs
    update(memory.sample(indexes)) with the indices the sampling kernel draws: identical parameters and losses.
    mode: "ring" = the one-launch update draws and gathers its batch from the replay ring itself; "gathered" = a draw + gather
    launch in front of it; "layered" = the grouped-GEMM path."""
    from xuance_amd import ops
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    cfg = dict(representation_hidden_size=[64], q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32,
               activation="relu", seed=1, parallels=16, running_steps=10 ** 6, buffer_size=16 * 40, batch_size=32,
               learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05, decay_step_greedy=50000,
               sync_frequency=5, training_frequency=1, start_training=10 ** 9, n_epochs=4, use_grad_clip=True,
               grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True, use_rnn=False,
               distributed_training=False, device="cuda", model_dir="/tmp/x", use_fused_qmix_update=mode != "layered",
               fused_qmix_gather_in_kernel=mode == "ring")
    res = []
    for graph in (False, True):
        torch.manual_seed(0)
        np.random.seed(0)
        agent = QMIX_Agents(Namespace(**cfg), SyntheticSMACVecEnv(16, seed=3))
        agent.train(25)                                       # fills 25 ring rows, no updates (start_training is far away)
        assert agent.learner.fused_eligible() == (mode != "layered")
        lr, mem = agent.learner, agent.memory
        assert mem.size == 25 and int(mem.size_dev.item()) == 25
        infos = []

        def perturb(it):                                      # between phases somebody else writes the parameters: the weight
            if it == 1:                                       # images of the one-launch update must follow (model.version)
                agent.model.load_state_dict({k: v * 1.01 for k, v in agent.model.state_dict().items()})
        if graph:
            for it in range(3):                               # first call eager + capture, then two graph launches
                perturb(it)
                infos.append(lr.update_from_buffer(mem, 4, seed=7))
            assert lr._buf_graph is not None
        else:
            idx = torch.zeros(32, dtype=torch.int64, device="cuda")
            ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
            for it in range(3):
                perturb(it)
                for e in range(4):
                    ops.sample_replay_indices(idx, mem.n_envs, mem.n_size, mem.size_dev, 7, 0, ctr)
                    ops.counter_add(ctr, 1)
                    assert int(idx.max()) < 16 * mem.n_size and int((idx % mem.n_size).max()) < 25
                    info = lr.update(mem.sample(indexes=idx.clone()))
                infos.append(info)
        torch.cuda.synchronize()
        assert lr.iterations == 12
        res.append((agent.model.params.flat.cpu().numpy().copy(), agent.model.target_flat.cpu().numpy().copy(),
                    np.array([[i["loss_Q"], i["predictQ"]] for i in infos])))
    (pa, ta, ia), (pb, tb, ib) = res
    assert np.array_equal(pa, pb) and np.array_equal(ta, tb) and np.array_equal(ia, ib)


@pytest.mark.parametrize("atari", [False, True])
def test_dqn_graph_update_phase_equals_eager_updates_on_the_same_indices(atari):
    """DQN_Learner.update_from_buffer (device sampling + gather + update per hipGraph launch) vs update(**memory.sample(
    indexes)) with the indices the sampling kernel draws: identical parameters, target parameters and losses."""
    from xuance_amd import ops
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv, SyntheticAtariVecEnv
    n = 8
    cfg = dict(representation="Basic_MLP", representation_hidden_size=[64], q_hidden_size=[64], activation="relu", seed=1,
               parallels=n, running_steps=10 ** 6, buffer_size=n * 40, batch_size=16, learning_rate=1e-3, gamma=0.99,
               start_greedy=0.5, end_greedy=0.05, decay_step_greedy=10 ** 5, sync_frequency=3, training_frequency=1,
               start_training=10 ** 9, n_epochs=2, use_grad_clip=True, grad_clip_norm=0.5, use_obsnorm=False,
               use_rewnorm=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
    if atari:
        cfg.update(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                   q_hidden_size=[512])
    res = []
    for graph in (False, True):
        torch.manual_seed(0)
        np.random.seed(0)
        env = SyntheticAtariVecEnv(n, seed=2) if atari else DeviceCartPoleVecEnv(n, seed=1)
        agent = DQN_Agent(Namespace(**cfg), env)
        agent.train(20)                                       # fills 20 ring rows, no updates
        lr, mem = agent.learner, agent.memory
        assert mem.size == 20 and int(mem.size_dev.item()) == 20
        infos = []
        if graph:
            for _ in range(3):
                infos.append(lr.update_from_buffer(mem, 2, seed=5))
            assert lr._buf_graph is not None
        else:
            idx = torch.zeros(16, dtype=torch.int64, device="cuda")
            ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
            for _ in range(3):
                for e in range(2):
                    ops.sample_replay_indices(idx, mem.n_envs, mem.n_size, mem.size_dev, 5, 0, ctr)
                    ops.counter_add(ctr, 1)
                    info = lr.update(**mem.sample(indexes=idx.clone()))
                infos.append(info)
        torch.cuda.synchronize()
        assert lr.iterations == 6
        res.append((agent.model.params.flat.cpu().numpy().copy(), agent.model.target_flat.cpu().numpy().copy(),
                    np.array([[i["Qloss"], i["predictQ"]] for i in infos])))
    (pa, ta, ia), (pb, tb, ib) = res
    assert np.abs(pa).max() > 0 and np.array_equal(pa, pb) and np.array_equal(ta, tb) and np.array_equal(ia, ib)


def test_dqn_cnn_acting_in_one_launch_equals_the_layered_acting():
    """DQN_Agent's acting step on a convolutional Q network: convolutions, then pool + hidden + Q layers + the epsilon-greedy choice
    as ONE launch (xrl_dqn_act_tail) against the layered launches (max-pool, GEMM, Q layer, xrl_egreedy): same Philox coin and random
    action per env, Q values within 1e-5 of their scale (the hidden layer's sum runs in another order), so the same actions unless
    two Q values of an env tie within that -- every stored action is compared; and the same loop with updates runs on."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n = 16
    cfg = dict(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
               q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 6, buffer_size=n * 64, batch_size=16,
               learning_rate=1e-3, gamma=0.99, start_greedy=0.3, end_greedy=0.05, decay_step_greedy=10 ** 5, sync_frequency=3,
               training_frequency=1, start_training=10 ** 9, n_epochs=1, use_grad_clip=False, use_obsnorm=False, use_rewnorm=False,
               distributed_training=False, device="cuda", model_dir="/tmp/x")
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        agent = DQN_Agent(Namespace(use_fused_q_tail=fused, **cfg), SyntheticAtariVecEnv(n, seed=2))
        assert agent._act_fused == fused
        agent.train(24)
        torch.cuda.synchronize()
        q = agent.model.plan.acts[2][:n, :4].cpu().numpy().copy()
        res.append((agent.memory.soa.fields["actions"].cpu().numpy().copy() if hasattr(agent.memory, "soa") else
                    agent.memory.data["actions"].cpu().numpy().copy(), q, agent.envs.action.cpu().numpy().copy()))
    (aa, qa, ea), (ab, qb, eb) = res
    assert_close(qa, qb, 1e-5, "Q values of the last acting step")
    assert np.array_equal(aa, ab) and np.array_equal(ea, eb) and len(np.unique(aa)) > 1


def test_dqn_cnn_weight_images_kept_by_the_optimiser_launch_equal_rebuilt_ones():
    """The convolution stack's fragment-ordered weight images: kept current by xrl_reduce_adam's mirrors (and the target's image by
    the in-launch hard update) against rebuilt by xrl_gather_images in every pass (config.use_live_weight_images: False) -- one
    update per captured phase (the draw-counter tick and the loss sums ride in the optimiser launch as well), acting in between,
    and writers the optimiser does not see: load_state_dict and copy_target bump the network's version, the next phase re-packs
    before its graph runs.  Bit-equal parameters, targets, losses and stored actions."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n = 8
    cfg = dict(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
               q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 6, buffer_size=n * 64, batch_size=16,
               learning_rate=1e-3, gamma=0.99, start_greedy=0.3, end_greedy=0.05, decay_step_greedy=10 ** 5, sync_frequency=3,
               training_frequency=1, start_training=10 ** 9, n_epochs=1, use_grad_clip=False, use_obsnorm=False, use_rewnorm=False,
               distributed_training=False, device="cuda", model_dir="/tmp/x")
    res = []
    for live in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        agent = DQN_Agent(Namespace(use_live_weight_images=live, **cfg), SyntheticAtariVecEnv(n, seed=2))
        agent.train(20)
        lr, mem, net = agent.learner, agent.memory, agent.model
        infos = [lr.update_from_buffer(mem, 1, seed=5) for _ in range(4)]
        assert lr._buf_graph is not None and net.conv.is_live() == live and net.conv.is_live(net.target_flat) == live
        agent.train(6)                                       # acting on the current weights (no updates: start_training)
        sd = {k: v * 0.5 for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        assert not net.conv.is_live()
        infos += [lr.update_from_buffer(mem, 1, seed=5) for _ in range(2)]
        net.copy_target()
        infos += [lr.update_from_buffer(mem, 1, seed=5) for _ in range(2)]
        agent.train(3)
        torch.cuda.synchronize()
        assert lr.iterations == 8
        res.append((net.params.flat.cpu().numpy().copy(), net.target_flat.cpu().numpy().copy(),
                    np.array([[i["Qloss"], i["predictQ"]] for i in infos]), mem.soa.fields["actions"].cpu().numpy().copy()
                    if hasattr(mem, "soa") else None))
    (pa, ta, ia, aa), (pb, tb, ib, ab) = res
    assert np.abs(pa).max() > 0 and np.array_equal(pa, pb) and np.array_equal(ta, tb) and np.array_equal(ia, ib)
    assert aa is None or np.array_equal(aa, ab)


def _rnn_cfg(**kw):
    c = dict(q_hidden_size=[64], fc_hidden_sizes=[64], recurrent_hidden_size=64, hidden_dim_mixing_net=32,
             hidden_dim_hyper_net=32, activation="relu", seed=1, parallels=8, running_steps=10 ** 6, buffer_size=64,
             batch_size=8, learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05,
             decay_step_greedy=50000, sync_frequency=3, training_frequency=1, start_training=0, n_epochs=2,
             use_grad_clip=True, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True, use_rnn=True,
             rnn_backprop_agents=True, episode_length=12, distributed_training=False, device="cuda", model_dir="/tmp/x",
             use_hip_graph=True)
    c.update(kw)
    return Namespace(**c)


def test_qmix_rnn_agents_episode_loop_vs_oracle(oracle):
    """Recurrent QMIX agents end to end on the device: run_episodes (GRU state carried and reset per env,
    staging rows -> episode ring), then update phases from the ring (device sampling + time-major gather + BPTT update
    as one hipGraph).  Checked: the carried GRU state against the oracle's GRU over the staged observations of the
    running episodes; every stored action was available; the update phases against the oracle replaying the same
    sampled episodes (the device's indices are the fixed input)."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    n, N, T = 8, 3, 12
    env = SyntheticSMACVecEnv(n, seed=3, max_episode_steps=T)
    # reference_rnn_reset False: a finished env restarts ITS OWN recurrent rows from zero, which is what check (1) restates; the
    # reference's rule (flattened row i, the default: tests/test_gpu_agent_replay.py replays it against the reference's own run)
    # leaves a finished env's rows running
    agent = QMIX_Agents(_rnn_cfg(reference_rnn_reset=False), env)
    mem, lr, net = agent.memory, agent.learner, agent.model
    sd = {k: v.cpu().numpy().copy() for k, v in net.state_dict().items()}
    agent.run_episodes(n)
    torch.cuda.synchronize()
    size = mem.size
    assert size >= n and agent.current_step > 0
    # (1) carried state: for every env the GRU over the staged observations of its running episode, from zeros
    steps = env.steps.cpu().numpy()
    obs_stage = mem.episode_data["obs"].cpu().numpy().reshape(n, T + 1, N, -1)
    h = agent.rnn_h.cpu().numpy().reshape(n, N, -1)
    for e in range(n):
        if steps[e] == 0:
            continue
        _, c = oracle.qmix_rnn_agent_forward(sd, "individual_q_networks.shared", obs_stage[e, :steps[e]].transpose(1, 0, 2))
        assert_close(h[e], c["hs"][:, -1], 1e-5, f"carried GRU state of env {e}")
    # (2) ring contents: filled prefix, available actions
    data = {k: v.cpu().numpy() for k, v in mem.data.items()}
    for ep in range(size):
        L = int(data["filled"][ep].sum())
        assert L >= 1 and data["filled"][ep, :L, 0].all()
        acts = data["actions"][ep, :L].astype(int)                                   # [L, N]
        av = data["avail_actions"][ep, :L].reshape(L, N, -1)
        assert np.take_along_axis(av, acts[..., None], -1).all()
    # (3) update phases (eager+capture, then graph replays) vs the oracle on the same episodes
    opt = oracle.AdamOracle({k: sd[k] for k in net.ref_order}, lr=7e-4, eps=1e-5, total_iters=lr.total_iters)
    chain, sd0 = ChainCheck(7e-4, total_iters=lr.total_iters), {k: v.copy() for k, v in sd.items()}
    cfg = dict(gamma=0.99, double_q=True, use_actions_mask=True, agent_grad=True)
    seen = []
    lr.callback.on_update_end = lambda it, **kw: seen.append(it) or {}
    for phase in range(3):
        info = lr.update_from_buffer(mem, 2, seed=5)
        # replay: the indices of the phase's epochs come from the sampling kernel with counter = 2*phase + e
        from xuance_amd import ops
        idx = torch.zeros(8, dtype=torch.int64, device="cuda")
        for e in range(2):
            ops.sample_replay_indices(idx, 1, mem.buffer_size, mem.size_dev, 5, 2 * phase + e, None)
            ii = idx.cpu().numpy()
            b = dict(obs=data["obs"][ii].reshape(8, T + 1, N, -1).transpose(0, 2, 1, 3),
                     actions=data["actions"][ii].transpose(0, 2, 1), rewards=data["rewards"][ii].transpose(0, 2, 1),
                     terminals=data["terminals"][ii].transpose(0, 2, 1), agent_mask=data["agent_mask"][ii].transpose(0, 2, 1),
                     avail_actions=data["avail_actions"][ii].reshape(8, T + 1, N, -1).transpose(0, 2, 1, 3),
                     state=data["state"][ii], filled=data["filled"][ii][..., 0])
            oi, grads = oracle.qmix_rnn_forward_backward(sd, b, cfg)
            oracle.AdamOracle.clip_grad_norm_(grads, 10.0)
            opt.step(grads)
            chain.step(grads)
            if (2 * phase + e + 1) % 3 == 0:
                oracle.qmix_copy_target(sd)
        assert_close(info["loss_Q"], oi["loss"], 1e-5, "loss_Q")
        # parameters: within what gradients agreeing at 1e-5 of their scale allow (conftest.ChainCheck: an Adam step on an entry
        # whose gradient sits at eps is ill-conditioned -- a 2-update-old GRU bias moved 2.6e-5 of its 1.4e-3 apart on one box)
        got = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
        chain.check({k: got[k] for k in chain.allow}, {k: sd[k] for k in chain.allow}, sd0, what=f"phase {phase} param")
        ev = lambda k: k[len("target_"):] if k.startswith("target_individual") else "eval_Qtot." + k[len("target_Qtot."):]
        tk = [k for k in sd if k.startswith("target_") and ev(k) in chain.allow]
        chain.check({ev(k): got[k] for k in tk}, {ev(k): sd[k] for k in tk}, sd0, what=f"phase {phase} target copy of")
    assert lr._buf_graph is not None and lr.iterations == 6 and seen == [1, 2, 3, 4, 5, 6]


def test_qmix_rnn_agents_train_at_3m_shape():
    """configs/qmix/sc2/3m.yaml shapes and defaults (64 envs here, 60-step episodes, batch 32 episodes, n_epochs 8,
    reference behaviour: agents' gradient detached): the whole train() loop."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    env = SyntheticSMACVecEnv(64, seed=3)
    agent = QMIX_Agents(_rnn_cfg(parallels=64, buffer_size=5000, batch_size=32, n_epochs=8, sync_frequency=200,
                                 start_training=1000, use_grad_clip=False, rnn_backprop_agents=False, episode_length=60), env)
    a0 = agent.model.params.view(agent.model.w_hh).clone()
    m0 = agent.model.params.view("eval_Qtot.hyper_b_1.weight").clone()
    info = agent.train(120)
    assert agent.current_step >= 120 * 64 and agent.learner.iterations > 0 and agent.learner.iterations % 8 == 0
    assert np.isfinite(info["loss_Q"]) and np.isfinite(info["predictQ"])
    assert torch.equal(agent.model.params.view(agent.model.w_hh), a0)              # iql_learner.py:49,58: agents detached
    assert not torch.equal(agent.model.params.view("eval_Qtot.hyper_b_1.weight"), m0)


def test_perdqn_agent_loop():
    """PerDQN_Agent (perdqn_agent.py:12-96): prioritized sampling / update / priority refresh every training step on the
    Atari-shaped provider; priorities of sampled transitions become |td|^alpha, unsampled ones keep max_priority^alpha."""
    import random
    from xuance_amd.agents import PerDQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    n = 16
    cfg = Namespace(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                    q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 6, buffer_size=n * 32,
                    batch_size=32, learning_rate=1e-4, gamma=0.99, start_greedy=0.5, end_greedy=0.05, decay_step_greedy=10 ** 6,
                    sync_frequency=500, training_frequency=n, start_training=n * 4, use_grad_clip=False, grad_clip_norm=0.5,
                    use_obsnorm=False, use_rewnorm=False, PER_alpha=0.6, PER_beta0=0.4, use_hip_graph=False,
                    distributed_training=False, device="cuda", model_dir="/tmp/x")
    agent = PerDQN_Agent(cfg, SyntheticAtariVecEnv(n, seed=2))
    info = agent.train(12)
    mem = agent.memory
    assert agent.learner.iterations == 7 and np.isfinite(info["Qloss"])
    assert abs(agent.PER_beta - (0.4 + 7 * 0.6 / 12)) < 1e-12
    leaves = mem.it_sum[:, mem.capacity:mem.capacity + mem.size].cpu().numpy()
    mp = mem.max_priority.cpu().numpy()
    assert (leaves > 0).all() and (mp >= 1.0).all()
    # root == sum of leaves, min tree root == min of leaves (float64, same association order per level is not required here)
    assert_close(mem.it_sum[:, 1].cpu().numpy(), leaves.sum(1), 1e-12, "sum tree root")
    assert_close(mem.it_min[:, 1].cpu().numpy(), leaves.min(1), 0.0, "min tree root")
    # the most recently stored step has not been sampled-and-updated after its store only if it still carries max_priority^alpha
    last = leaves[:, (mem.ptr - 1) % mem.n_size]
    assert ((np.abs(last - mp ** 0.6) < 1e-12) | (last != mp ** 0.6)).all()


def test_qmix_lstm_agents_episode_loop(oracle):
    """`rnn: "LSTM"`: hidden AND cell state of every (env, agent) row are carried between acting steps and zeroed with the
    env's episode; checked against the oracle's LSTM over the staged observations of the running episodes."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    n, N, T = 8, 3, 12
    for seed in range(3, 20):          # a provider seed for which some episode ends early, so that envs are mid-episode at the end
        env = SyntheticSMACVecEnv(n, seed=seed, max_episode_steps=T)
        agent = QMIX_Agents(_rnn_cfg(rnn="LSTM", reference_rnn_reset=False), env)      # (the env's own rows restart: see the GRU twin above)
        assert agent.model.lstm and agent.model.G == 256
        sd = {k: v.cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
        agent.run_episodes(n)
        torch.cuda.synchronize()
        steps = env.steps.cpu().numpy()
        if (steps > 0).any():
            break
    obs_stage = agent.memory.episode_data["obs"].cpu().numpy().reshape(n, T + 1, N, -1)
    h, c = agent.rnn_h.cpu().numpy().reshape(n, N, -1), agent.rnn_c.cpu().numpy().reshape(n, N, -1)
    checked = 0
    for e in range(n):
        if steps[e] == 0:
            continue
        _, cc = oracle.qmix_rnn_agent_forward(sd, "individual_q_networks.shared", obs_stage[e, :steps[e]].transpose(1, 0, 2))
        assert_close(h[e], cc["hs"][:, -1], 1e-5, f"carried h of env {e}")
        assert_close(c[e], cc["gru"]["cs"][:, -1], 1e-5, f"carried c of env {e}")
        checked += 1
    assert checked > 0
    info = agent.learner.update_from_buffer(agent.memory, 2, seed=5)
    assert np.isfinite(info["loss_Q"])


@pytest.mark.parametrize("name", ["DDQN", "Duel_DQN"])
def test_sibling_dqn_agents_from_the_registry(name):
    """REGISTRY_Agents["DDQN"] / ["Duel_DQN"] (reference keys, agents/__init__.py:84-86): the DQN loop with the sibling
    learner / the dueling network; greedy actions of the dueling head = argmax of V + A - mean(A)."""
    from xuance_amd.agents import REGISTRY_Agents
    from xuance_amd.envs import DeviceCartPoleVecEnv
    from xuance_amd.learners import DDQN_Learner, DuelDQN_Learner
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[64], q_hidden_size=[64], activation="relu", seed=1,
                    parallels=16, running_steps=200000, buffer_size=16 * 500, batch_size=64, learning_rate=1e-3, gamma=0.99,
                    start_greedy=0.0, end_greedy=0.0, decay_step_greedy=20000, sync_frequency=50, training_frequency=16,
                    start_training=200, use_grad_clip=False, grad_clip_norm=0.5, use_obsnorm=False, use_rewnorm=False,
                    distributed_training=False, device="cuda", model_dir="/tmp/x")
    env = DeviceCartPoleVecEnv(16, seed=1)
    agent = REGISTRY_Agents[name](cfg, env)
    assert isinstance(agent.learner, DDQN_Learner if name == "DDQN" else DuelDQN_Learner)
    assert agent.model.dueling == (name == "Duel_DQN")
    p0 = agent.model.params.flat.clone()
    info = agent.train(60)
    assert agent.learner.iterations > 0 and np.isfinite(info["Qloss"])
    assert float((agent.model.params.flat - p0).abs().max()) > 0
    # epsilon = 0: the stored action of the last step is the greedy one for the observation it was taken on
    n, A = 16, 2
    t = (agent.memory.ptr - 1) % agent.memory.n_size
    obs = agent.memory.soa.fields["observations"][t].clone()
    act = agent.memory.soa.fields["actions"][t].cpu().numpy()
    sd = {k: v.cpu().numpy() for k, v in agent.model.state_dict().items()}
    out = agent.model.forward(obs.view(n, -1), n)[:n].cpu().numpy()
    if name == "Duel_DQN":
        q = out[:, A:A + 1] + out[:, :A] - out[:, :A].mean(1, keepdims=True)          # q_head.py:77
    else:
        q = out[:, :A]
    # the last update ran AFTER the last act pass: compare only where the greedy choice is not a near-tie
    gap = np.abs(q[:, 0] - q[:, 1])
    ok = gap > 1e-2
    assert ok.any() and np.array_equal(act[ok], q.argmax(1)[ok].astype(np.float32))


def test_runner_surface_of_the_dqn_and_qmix_agents(tmp_path):
    """save_model / load_model / test / get_actions / meta_data on the off-policy agents (run_drl.py:101-203;
    core/off_policy.py:150-171, 272-350; core/off_policy_marl.py:596-640), with host evaluation envs."""
    from argparse import Namespace
    from xuance_amd.agents import DQN_Agent, QMIX_Agents
    from xuance_amd.envs import DeviceCartPoleVecEnv, DummyVecEnv, NumpyCartPoleEnv, DummyVecMultiAgentEnv, HostSMACLikeEnv, \
        SyntheticSMACVecEnv
    from oracle import xrl_oracle as o
    base = dict(seed=1, gamma=0.99, learning_rate=1e-3, start_greedy=0.5, end_greedy=0.05, decay_step_greedy=10000,
                sync_frequency=50, training_frequency=1, running_steps=100000, use_grad_clip=False, grad_clip_norm=0.5,
                distributed_training=False, device="cuda", use_obsnorm=True, use_rewnorm=False, obsnorm_range=5, rewnorm_range=5)
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[64], q_hidden_size=[64], activation="relu",
                    parallels=16, buffer_size=16 * 64, batch_size=32, start_training=64, model_dir=str(tmp_path / "dqn"),
                    agent="DQN", **base)
    a = DQN_Agent(cfg, DeviceCartPoleVecEnv(16, seed=2))
    a.train(40)
    a.save_model("final_train_model.pth")
    b = DQN_Agent(Namespace(**vars(cfg)), DeviceCartPoleVecEnv(16, seed=2))
    b.load_model(b.model_dir_load)
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[k]), k
    assert torch.equal(a.obs_mean, b.obs_mean) and b.meta_data["algo"] == "DQN"
    scores = b.test(test_episodes=5, test_envs=DummyVecEnv([NumpyCartPoleEnv] * 3, env_seed=4))
    assert len(scores) >= 5 and all(s >= 8 for s in scores)
    obs = np.random.default_rng(1).standard_normal((7, 4)).astype(np.float32)
    sd = {k: v.cpu().numpy() for k, v in b.model.state_dict().items()}
    q = obs
    for name in ("representation.model.0", "eval_Q_head.q_value.0"):
        q = np.maximum(q @ sd[name + ".weight"].T + sd[name + ".bias"], 0)
    q = q @ sd["eval_Q_head.q_value.2.weight"].T + sd["eval_Q_head.q_value.2.bias"]
    assert np.array_equal(b.get_actions(obs, test_mode=True).env_actions, q.argmax(-1))
    # ---- QMIX (recurrent, the 3m default) on a host multi-agent env
    mcfg = Namespace(use_rnn=True, rnn="GRU", fc_hidden_sizes=[64], recurrent_hidden_size=64, q_hidden_size=[64], activation="relu",
                     hidden_dim_mixing_net=32, hidden_dim_hyper_net=32, parallels=8, buffer_size=64, batch_size=8,
                     start_training=0, n_epochs=2, double_q=True, use_actions_mask=True, use_parameter_sharing=True,
                     model_dir=str(tmp_path / "qmix"), agent="QMIX", **{**base, "use_obsnorm": False, "use_grad_clip": True,
                                                                        "grad_clip_norm": 10.0})
    env = SyntheticSMACVecEnv(8, seed=3, max_episode_steps=12, p_term=0.05)
    m = QMIX_Agents(mcfg, env)
    m.train(12)
    m.save_model("final_train_model.pth")
    m2 = QMIX_Agents(Namespace(**vars(mcfg)), SyntheticSMACVecEnv(8, seed=3, max_episode_steps=12, p_term=0.05))
    m2.load_model(m2.model_dir_load)
    for k, v in m.model.state_dict().items():
        assert torch.equal(v, m2.model.state_dict()[k]), k
    tenv = DummyVecMultiAgentEnv([HostSMACLikeEnv] * 4, env_seed=9)
    scores = m2.test(test_episodes=5, test_envs=tenv, close_envs=True)       # (an unavailable action would assert in the env)
    assert len(scores) >= 5 and all(0.0 <= s <= 60.0 for s in scores) and tenv.closed
    assert "Test-Results/Episode-Rewards" in m2.logged[-1][1]


@pytest.mark.parametrize("lag,unroll,merged", [(0, 2, False), (1, 4, False), (3, 2, False), (1, 8, False), (1, 4, True)])
def test_captured_vector_step_equals_the_eager_episode_loop(lag, unroll, merged):
    """run_episodes of the recurrent QMIX agents with the vector step captured as one hipGraph per observation-buffer set
    (use_hip_graph; `unroll` steps per graph launch) vs the eager launch sequence: the same Philox step indices (device counters that start at the host's
    values), hence the same actions, episodes, ring contents, exploration schedule and step accounting -- also when the
    host enqueues `lag` graph launches ahead of its knowledge of the loop condition (the dry steps after the call's last
    episode change nothing a later call or an update can see; the GRU state and the episode staging they touch are
    re-zeroed by the next call and not compared).  merged (round 6, config.gate_in_finish; measured slower, off by default): the loop's gate rides in the store + finish launch
    (xrl_episode_store_finish_gate: the last block of the finish carries it) instead of a launch of its own."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    res = []
    for graph in (False, True):
        torch.manual_seed(0)
        agent = QMIX_Agents(_rnn_cfg(use_hip_graph=graph, start_training=10 ** 9, start_greedy=0.6, end_greedy=0.05,
                                     decay_step_greedy=400, episode_loop_lag=lag, episode_loop_unroll=unroll, gate_in_finish=merged),
                            SyntheticSMACVecEnv(8, seed=3, max_episode_steps=12, p_term=0.05))
        for _ in range(3):
            agent.run_episodes(8)
        torch.cuda.synchronize()
        assert bool(getattr(agent, "_steps_g", None)) == graph
        mem = agent.memory
        res.append(dict(ptr_size=mem.ptr_size.cpu().numpy(), step=np.array([agent.current_step, agent._host_step, agent.envs._host_step]),
                        eps=np.array([agent.e_greedy]), eps_dev=agent.eps_dev.cpu().numpy(),

                        **{k: v.cpu().numpy() for k, v in mem.data.items()},
                        ))
    a, b = res
    assert a["ptr_size"][1] >= 24 and a["step"][0] > 0
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_one_launch_acting_step_of_the_feed_forward_agents_vs_the_layered_path():
    """xrl_marl_act_gru with H = 0 (no recurrent layer: Basic_MLP 64 + Q head 64-9 of the 3m shape) incl. the action selection
    against three xrl_linear_fwd launches + xrl_marl_select_actions: Q values 1e-5, actions identical."""
    from xuance_amd.nets import MixingQNet
    torch.manual_seed(0)
    net = MixingQNet(3, 30, 9, 48, (64,), (64,), 32, 32, "relu")
    assert net.act_image() is not None and net.act_image().struct.H == 0
    g = torch.Generator(device="cpu").manual_seed(2)
    for R in (192, 50):
        X = torch.randn(R, 30, generator=g).cuda()
        avail = (torch.rand(R, 9, generator=g) < 0.6).float()
        avail[:, 0] = 1
        acts = [torch.zeros(R, dtype=torch.int32, device="cuda") for _ in range(2)]
        sel = lambda a: dict(avail=avail.cuda(), eps_dev=torch.tensor([0.3], device="cuda"), action=a, action_f=None, seed=9,
                             step=R, step_dev=None)
        q_a = net.act_step(X, R, None, fused=True, select=sel(acts[0])).clone()
        q_b = net.act_step(X, R, None, fused=False, select=sel(acts[1])).clone()
        assert_close(q_a.cpu().numpy()[:R, :9], q_b.cpu().numpy()[:R, :9], 1e-5, "q")     # (the plan's buffer may be larger)
        assert np.array_equal(acts[0].cpu().numpy(), acts[1].cpu().numpy())
        assert (avail.numpy()[np.arange(R), acts[0].cpu().numpy()] == 1).all()


@pytest.mark.parametrize("R", [192, 7, 24])
def test_one_launch_acting_step_of_the_recurrent_agents_vs_the_layered_path(R):
    """xrl_marl_act_gru (mlp block -> GRU cell -> Q head from one LDS-staged weight image; rows split over workgroups, a
    ragged last group) against xrl_linear_fwd + xrl_gru_forward + xrl_linear_fwd on the same observations: Q values and
    carried hidden states over several steps with row resets, 1e-5; and after a parameter change + refresh()."""
    from xuance_amd.nets import MixingQNet
    torch.manual_seed(0)
    net = MixingQNet(3, 30, 9, 48, (), (64,), 32, 32, "relu", use_rnn=True, fc_hidden=(64,), recurrent_hidden=64)
    assert net.act_image() is not None
    g = torch.Generator(device="cpu").manual_seed(1)
    h_a, h_b = torch.zeros(R, 64, device="cuda"), torch.zeros(R, 64, device="cuda")
    for step in range(5):
        if step == 3:                                              # parameters move: the image must follow on refresh()
            net.params.flat.add_(0.01 * torch.randn(net.params.flat.shape, generator=g).cuda())
            net.act_image().refresh()
        X = torch.randn(R, 30, generator=g).cuda()
        reset = (torch.rand(R, generator=g) < (0.3 if step else 0.0)).float().cuda()
        avail = (torch.rand(R, 9, generator=g) < 0.6).float()
        avail[:, 0] = 1
        acts = [torch.zeros(R, dtype=torch.int32, device="cuda") for _ in range(2)]
        sel = lambda a: dict(avail=avail.cuda(), eps_dev=torch.tensor([0.4], device="cuda"), action=a, action_f=None, seed=5,
                             step=step, step_dev=None)     # (the selection in the same launch vs xrl_marl_select_actions)
        q_a = net.act_step(X, R, h_a, reset, fused=True, select=sel(acts[0])).clone()
        q_b = net.act_step(X, R, h_b, reset, fused=False, select=sel(acts[1])).clone()
        assert_close(q_a.cpu().numpy(), q_b.cpu().numpy(), 1e-5, f"q step {step}")
        assert np.array_equal(acts[0].cpu().numpy(), acts[1].cpu().numpy()), f"actions step {step}"
        assert (avail.numpy()[np.arange(R), acts[0].cpu().numpy()] == 1).all()
        assert_close(h_a.cpu().numpy(), h_b.cpu().numpy(), 1e-5, f"h step {step}")
    assert float(h_a.abs().max()) > 0


@pytest.mark.parametrize("rnn", [True, False])
@pytest.mark.parametrize("R,rpw", [(192, 6), (192, 1), (7, 4), (50, 3)])
def test_acting_launch_with_weights_in_registers_is_bit_identical_to_the_lds_staged_form(rnn, R, rpw):
    """xrl_marl_act_gru's two kernels (round 6): one thread per output with its weight row in registers (marl_act_rows_kernel, the
    default where the shape allows it) vs the weight image staged in LDS (marl_act_gru_kernel): every output is the same fma chain in
    the same order, so Q values, carried hidden states and selected actions must be EQUAL -- over several steps with row resets, a
    ragged last workgroup, recurrent and feed-forward agents."""
    import ctypes as C
    from xuance_amd import ops, _lib
    from xuance_amd.nets import MixingQNet
    torch.manual_seed(0)
    if rnn:
        net = MixingQNet(3, 30, 9, 48, (), (64,), 32, 32, "relu", use_rnn=True, fc_hidden=(64,), recurrent_hidden=64)
    else:
        net = MixingQNet(3, 30, 9, 48, (64,), (64,), 32, 32, "relu")
    sts = [ops.MarlActGruState(net, rows_per_wg=1, lds_staged=False), ops.MarlActGruState(net, rows_per_wg=rpw, lds_staged=True)]
    im = _lib.QaImage()
    for st, inter in zip(sts, (1, 0)):
        _lib.call("xrl_marl_act_gru_layout", C.byref(st.struct), C.byref(im))
        assert im.interleaved == inter
    g = torch.Generator(device="cpu").manual_seed(4)
    hs = [torch.zeros(R, 64, device="cuda") for _ in range(2)] if rnn else [None, None]
    if True:
        for step in range(4):
            X = torch.randn(R, 30, generator=g).cuda()
            reset = (torch.rand(R, generator=g) < (0.3 if step else 0.0)).float().cuda() if rnn else None
            avail = (torch.rand(R, 9, generator=g) < 0.6).float()
            avail[:, 0] = 1
            avail = avail.cuda()
            eps = torch.tensor([0.4], device="cuda")
            outs = []
            for form in (0, 1):
                q = torch.full((R, 9), -7.0, device="cuda")
                a, af = torch.zeros(R, dtype=torch.int32, device="cuda"), torch.zeros(R, device="cuda")
                sts[form].launch(X, R, hs[form], reset, q, select=dict(avail=avail, eps_dev=eps, action=a, action_f=af, seed=5, step=step, step_dev=None))
                torch.cuda.synchronize()
                outs.append((q, a, af))
            assert torch.equal(outs[0][0], outs[1][0]), f"q step {step}"
            assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), f"actions step {step}"
            if rnn:
                assert torch.equal(hs[0], hs[1]), f"h step {step}"
            assert float(outs[0][0].abs().max()) > 0 and (avail.cpu().numpy()[np.arange(R), outs[0][1].cpu().numpy()] == 1).all()


def test_shm_multi_agent_vec_env_device_path_and_evaluation(tmp_path):
    """ShmSubprocVecMultiAgentEnv on the GPU box: the shared block is page-locked, step_to_device lands a whole vector step
    in HBM with one copy (device tensors equal the host views), a worker failure raises instead of hanging, and
    QMIX_Agents.test() evaluates on it exactly as on DummyVecMultiAgentEnv (same seeds -> same scores)."""
    from argparse import Namespace
    from xuance_amd.envs import ShmSubprocVecMultiAgentEnv, DummyVecMultiAgentEnv, HostSMACLikeEnv, SyntheticSMACVecEnv
    from xuance_amd.agents import QMIX_Agents
    venv = ShmSubprocVecMultiAgentEnv([HostSMACLikeEnv] * 4, env_seed=9, in_series=2)       # (workers forked before any launch below)
    venv2 = ShmSubprocVecMultiAgentEnv([HostSMACLikeEnv] * 4, env_seed=9, in_series=2)
    assert venv._pinned
    venv.reset()
    acts = np.array([[int(np.flatnonzero(venv.buf_avail_actions[e][a])[0]) for a in venv.agents] for e in range(4)])
    dev = venv.step_to_device(torch.as_tensor(acts, device="cuda"))
    torch.cuda.synchronize()
    for k, t in dev.items():
        assert np.array_equal(t.cpu().numpy(), venv.v[k]), k
    assert tuple(dev["obs"].shape) == (4, 3, 30) and tuple(dev["avail"].shape) == (4, 3, 9) and tuple(dev["state"].shape) == (4, 48)
    with pytest.raises(RuntimeError, match="unavailable action"):    # strict env: an unavailable action is an error in the worker
        for _ in range(50):
            venv.step(np.full((4, 3), 8))
    venv.closed = True                                               # (its workers are gone)
    cfg = Namespace(use_rnn=True, rnn="GRU", fc_hidden_sizes=[64], recurrent_hidden_size=64, q_hidden_size=[64], activation="relu",
                    hidden_dim_mixing_net=32, hidden_dim_hyper_net=32, parallels=8, buffer_size=64, batch_size=8, start_training=0,
                    n_epochs=2, double_q=True, use_actions_mask=True, use_parameter_sharing=True, model_dir=str(tmp_path / "q"),
                    agent="QMIX", seed=1, gamma=0.99, learning_rate=1e-3, start_greedy=0.5, end_greedy=0.05,
                    decay_step_greedy=10000, sync_frequency=50, training_frequency=1, running_steps=100000, use_grad_clip=True,
                    grad_clip_norm=10.0, distributed_training=False, device="cuda")
    torch.manual_seed(0)
    m = QMIX_Agents(cfg, SyntheticSMACVecEnv(8, seed=3, max_episode_steps=12, p_term=0.05))
    s_shm = m.test(test_episodes=6, test_envs=venv2, close_envs=True)
    s_ref = m.test(test_episodes=6, test_envs=DummyVecMultiAgentEnv([HostSMACLikeEnv] * 4, env_seed=9), close_envs=True)
    assert venv2.closed and len(s_shm) >= 6 and s_shm == s_ref


def test_qmix_phase_launch_is_bit_identical_to_the_launch_pairs():
    """xrl_qmix_fused_phase (round 6): the 8 updates of a feed-forward QMIX vector step as ONE launch -- resident workgroups, slab sums /
    Adam / target sync / weight images between two updates inside the launch -- against the captured sequence of 8 x {xrl_qmix_fused_update,
    xrl_reduce_adam} it replaces: two agents that differ in this switch only, same seeds, same device provider, 40 vector steps (a few
    hundred updates, several hard target syncs): parameters, target, moments, both weight images, the acting image, the replay ring and
    the optimiser's counters must be EQUAL (a stale weight or slab read between two workgroups of the launch would part the runs)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_secondary as bs
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    agents = []
    for phase in (True, False):
        cfg = bs._qmix_cfg(64, False)
        cfg.use_qmix_phase_launch = phase
        cfg.sync_frequency = 50
        torch.manual_seed(0)
        agents.append(QMIX_Agents(cfg, SyntheticSMACVecEnv(64, seed=3)))
    a, b = agents
    for rnd in range(4):
        a.train(12); b.train(12)
        torch.cuda.synchronize()
        la, lb = a.learner, b.learner
        assert la._phase_launch and not lb._phase_launch, "the switch did not select the two paths"
        assert int(la._phase_sync[2].item()) == 0, "a meeting of the phase launch timed out"
        sa, sb = la.read_optimizer(), lb.read_optimizer()
        assert sa.step == sb.step > 0 and sa.sched_steps == sb.sched_steps and sa.last_lr == sb.last_lr
        assert abs(sa.last_grad_norm - sb.last_grad_norm) <= 1e-12 * max(1.0, abs(sb.last_grad_norm))      # (reported only: float64 sum order)
        for name in ("flat",):
            assert torch.equal(a.model.params.flat, b.model.params.flat), f"round {rnd}: parameters"
        assert torch.equal(a.model.target_flat, b.model.target_flat), f"round {rnd}: target parameters"
        assert torch.equal(la.optimizer.m, lb.optimizer.m) and torch.equal(la.optimizer.v, lb.optimizer.v) and torch.equal(la.optimizer.grad, lb.optimizer.grad)
        assert torch.equal(la._fused.img_eval, lb._fused.img_eval) and torch.equal(la._fused.img_target, lb._fused.img_target)
        if getattr(a.model, "_act_state", None) is not None:
            assert torch.equal(a.model._act_state.image, b.model._act_state.image), "acting launch's weight image"
        assert torch.equal(la._epoch_sums, lb._epoch_sums), "loss sums of the last phase"
        assert torch.equal(la._sample_counter, lb._sample_counter)
        for k, v in a.memory.soa.fields.items():
            assert torch.equal(v, b.memory.soa.fields[k]), f"round {rnd}: replay ring field {k}"
    assert sa.step >= 150 and sa.step // 50 >= 3, "the run did not cross several hard target syncs"


def test_conv_weight_gradients_of_all_layers_in_one_launch_equal_the_launch_per_kind():
    """conv_dw_mixed_kernel (round 6): the weight gradients of the uint8 first layer (32 filters) and of the float32 upper layers (64
    filters) as ONE launch against one launch per kind (xrl_set_conv_dw_mixed(0)): the same body per workgroup, so six DQN updates on the
    Atari-shaped network (captured phases, hard target syncs in between) must leave bit-equal parameters, targets and losses."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    from xuance_amd._lib import call
    n = 8
    cfg = dict(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
               q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 6, buffer_size=n * 64, batch_size=16,
               learning_rate=1e-3, gamma=0.99, start_greedy=0.3, end_greedy=0.05, decay_step_greedy=10 ** 5, sync_frequency=3,
               training_frequency=1, start_training=10 ** 9, n_epochs=1, use_grad_clip=False, use_obsnorm=False, use_rewnorm=False,
               distributed_training=False, device="cuda", model_dir="/tmp/x")
    res = []
    try:
        for mixed in (1, 0):
            call("xrl_set_conv_dw_mixed", mixed)
            torch.manual_seed(0)
            np.random.seed(0)
            agent = DQN_Agent(Namespace(**cfg), SyntheticAtariVecEnv(n, seed=2))
            agent.train(20)
            lr, mem, net = agent.learner, agent.memory, agent.model
            infos = [lr.update_from_buffer(mem, 1, seed=5) for _ in range(6)]
            torch.cuda.synchronize()
            res.append((net.params.flat.cpu().numpy().copy(), net.target_flat.cpu().numpy().copy(), np.array([[i["Qloss"], i["predictQ"]] for i in infos])))
    finally:
        call("xrl_set_conv_dw_mixed", 0)
    (pa, ta, ia), (pb, tb, ib) = res
    assert np.abs(pa).max() > 0 and np.array_equal(pa, pb) and np.array_equal(ta, tb) and np.array_equal(ia, ib)
