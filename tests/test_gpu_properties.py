"""GPU: size-independent properties of the hot-path kernels at BASELINE.json's FULL sizes, where replaying the oracle
element by element is too slow or pointless: permutation equivariance, round trips, closed forms, tree invariants,
determinism.  (Parity against the oracle / reference fixtures at small sizes lives in the other test_gpu_* files.)"""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    from xuance_amd import ops
    return ops


@pytest.mark.parametrize("n_envs,T", [(256, 256), (128, 256), (65536, 256)])          # C2, C4 per GPU, scale sweep
def test_gae_full_size_properties(ops, n_envs, T):
    """(1) envs are independent units: permuting the env columns permutes the outputs bit-exactly; (2) with gamma = lam = 1
    and no terminals the returns are the reverse cumulative sum of the rewards plus the bootstrap value (fp64 reference,
    fp32 tolerance scaled by the magnitude); (3) a terminal at step t makes everything before t independent of what
    follows it."""
    g = torch.Generator(device="cuda").manual_seed(1)
    rew = torch.randn(T, n_envs, device="cuda", generator=g)
    val = torch.randn(T, n_envs, device="cuda", generator=g)
    term = (torch.rand(T, n_envs, device="cuda", generator=g) < 0.02).float()
    bootv = torch.zeros(T, n_envs, device="cuda")
    bootv[T - 1] = torch.randn(n_envs, device="cuda", generator=g) * (1 - term[T - 1])
    seg = torch.zeros(T, n_envs, dtype=torch.uint8, device="cuda")
    seg[term > 0] = 3
    seg[T - 1] = torch.where(term[T - 1] > 0, 3, 1).to(torch.uint8)

    def run(r, v, d, b, s, gamma=0.98, lam=0.95):
        adv, ret = torch.zeros_like(r), torch.zeros_like(r)
        ops.gae_scan(r.contiguous(), v.contiguous(), d.contiguous(), b.contiguous(), s.contiguous(), adv, ret, gamma, lam, True)
        return adv, ret
    adv, ret = run(rew, val, term, bootv, seg)
    perm = torch.randperm(n_envs, device="cuda", generator=g)
    adv_p, ret_p = run(rew[:, perm], val[:, perm], term[:, perm], bootv[:, perm], seg[:, perm])
    assert torch.equal(adv_p, adv[:, perm]) and torch.equal(ret_p, ret[:, perm])
    # closed form
    zero = torch.zeros_like(term)
    seg1 = torch.zeros_like(seg); seg1[T - 1] = 1
    _, ret1 = run(rew, val, zero, bootv, seg1, 1.0, 1.0)
    ref = torch.flip(torch.cumsum(torch.flip(rew.double(), [0]), 0), [0]) + bootv[T - 1].double()
    scale = float(ref.abs().max())
    assert float((ret1.double() - ref).abs().max()) <= 2e-6 * scale * np.sqrt(T)
    # a terminal cuts the dependence
    rew2 = rew.clone(); rew2[T // 2 + 1:] += 5.0
    term2 = term.clone(); term2[T // 2] = 1.0
    seg2 = seg.clone(); seg2[T // 2] = 3
    a1, _ = run(rew, val, term2, bootv, seg2)
    a2, _ = run(rew2, val, term2, bootv, seg2)
    assert torch.equal(a1[:T // 2 + 1], a2[:T // 2 + 1])


def test_replay_ring_roundtrip_at_c3_size(ops):
    """C3: 64 envs of 84x84x4 uint8 frames (28 224 B rows): what store_step wrote is what gather returns, for every sampled
    (env, step), bit for bit, through the ring wrap."""
    from xuance_amd.memory import HipOffPolicyBuffer_Atari
    from xuance_amd.spaces import Box, Discrete
    n_envs, n_size, steps, bs = 64, 96, 130, 256
    buf = HipOffPolicyBuffer_Atari(Box(0, 255, (84, 84, 4), np.uint8), Discrete(4), None, n_envs, n_envs * n_size, bs)
    g = torch.Generator(device="cuda").manual_seed(2)
    # frame content encodes (step, env) so that any slot can be checked without keeping 1.7 GB of copies
    base = torch.randint(0, 256, (84 * 84 * 4,), dtype=torch.int32, device="cuda", generator=g)

    def frame(t, envs):
        return ((base[None, :] + (t * 7 + envs[:, None] * 13)) % 256).to(torch.uint8).view(-1, 84, 84, 4)
    envs = torch.arange(n_envs, device="cuda")
    for t in range(steps):
        buf.store(frame(t, envs), torch.full((n_envs,), float(t % 4), device="cuda"), torch.full((n_envs,), float(t), device="cuda"),
                  torch.zeros(n_envs, device="cuda"), frame(t + 1, envs))
    assert buf.size == n_size and buf.ptr == steps % n_size
    idx = torch.randint(0, n_envs * n_size, (bs,), device="cuda", generator=g)
    s = buf.sample(indexes=idx)
    env, slot = idx // n_size, idx % n_size
    t_written = torch.where(slot < steps % n_size, slot + (steps // n_size) * n_size, slot + (steps // n_size - 1) * n_size)
    for k in range(0, bs, 17):
        e, t = env[k:k + 1], int(t_written[k])
        assert torch.equal(s["obs"][k], frame(t, e)[0]) and torch.equal(s["obs_next"][k], frame(t + 1, e)[0])
    assert torch.equal(s["rewards"], t_written.float())


def test_episode_ring_roundtrip_at_3m_size(ops):
    """Recurrent QMIX buffer at configs/qmix/sc2/3m.yaml sizes (5 000 episodes x 61 slots x 3 agents): episodes finished in
    any order land in consecutive ring rows in env order; a time-major gather of sampled episodes returns them."""
    from xuance_amd.memory_marl import HipMARLOffPolicyBufferRNN
    from xuance_amd.spaces import Box, Discrete
    n, N, O, A, S, T, cap = 64, 3, 30, 9, 48, 60, 5000
    keys = [f"agent_{i}" for i in range(N)]
    buf = HipMARLOffPolicyBufferRNN(keys, Box(-1, 1, (S,)), {k: Box(-1, 1, (O,)) for k in keys}, {k: Discrete(A) for k in keys},
                                    n, cap, 32, T, use_actions_mask=True, avail_actions_shape={k: (A,) for k in keys})
    g = torch.Generator(device="cuda").manual_seed(3)
    steps = torch.zeros(n, dtype=torch.int32, device="cuda")
    ep_id = torch.arange(n, device="cuda").float()                 # identity of the episode each env is running
    next_id, stored = n, []
    for t in range(150):
        obs = ep_id[:, None, None] + 0.001 * steps[:, None, None].float() + torch.zeros(n, N, O, device="cuda")
        buf.store(obs=obs, actions=torch.zeros(n, N, device="cuda"), rewards=ep_id[:, None].expand(n, N).contiguous(),
                  terminals=torch.zeros(n, N, device="cuda"), agent_mask=torch.ones(n, N, device="cuda"),
                  avail_actions=torch.ones(n, N, A, device="cuda"), state=torch.zeros(n, S, device="cuda"), episode_steps=steps)
        done = (torch.rand(n, device="cuda", generator=g) < 0.05) | (steps + 1 >= T)
        buf.finish_paths(done.float(), steps + 1, obs=obs + 0.5, state=torch.zeros(n, S, device="cuda"),
                         avail_actions=torch.ones(n, N, A, device="cuda"))
        for e in torch.nonzero(done).flatten().tolist():            # env order == ring order
            stored.append((ep_id[e].item(), int(steps[e]) + 1))
            ep_id[e] = next_id; next_id += 1
        steps = torch.where(done, torch.zeros_like(steps), steps + 1)
    assert buf.size == len(stored) and buf.ptr == len(stored) % cap
    idx = torch.randint(0, len(stored), (32,), device="cuda", generator=g)
    dst = {"obs": torch.zeros(T + 1, 32, N * O, device="cuda"), "rewards": torch.zeros(T, 32, N, device="cuda"),
           "filled": torch.zeros(T, 32, 1, device="cuda")}
    buf.gather_into(idx, dst)
    for b, i in enumerate(idx.tolist()):
        eid, L = stored[i]
        assert float(dst["filled"][:, b, 0].sum()) == L and bool((dst["filled"][:L, b, 0] == 1).all())
        assert bool((dst["rewards"][:L, b] == eid).all())
        assert_close(dst["obs"][L - 1, b, 0].item(), eid + 0.001 * (L - 1), 1e-6, "last step obs")
        assert_close(dst["obs"][L, b, 0].item(), eid + 0.001 * (L - 1) + 0.5, 1e-6, "terminal obs slot")


def test_segment_tree_invariants_at_replay_size(ops):
    """Prioritized replay at 64 envs x 8 192 slots: after stores and many priority updates every internal node is exactly
    the sum (min) of its children, the root is the sum of the leaves, and the sampled steps follow the priorities."""
    from xuance_amd.memory import HipPerOffPolicyBuffer
    from xuance_amd.spaces import Box, Discrete
    n_envs, n_size, bs = 64, 8192, 64 * 64
    buf = HipPerOffPolicyBuffer(Box(-1, 1, (4,)), Discrete(2), None, n_envs, n_envs * n_size, bs, alpha=0.6)
    z = torch.zeros(n_envs, device="cuda")
    for t in range(300):
        buf.store(torch.zeros(n_envs, 4, device="cuda"), z, z, z, torch.zeros(n_envs, 4, device="cuda"))
    g = torch.Generator(device="cuda").manual_seed(4)
    for it in range(5):
        u = torch.rand(n_envs, buf.per_env, dtype=torch.float64, device="cuda", generator=g)
        s = buf.sample(0.5, uniforms=u.cpu().numpy())
        pr = torch.rand(n_envs, buf.per_env, device="cuda", generator=g) * (10.0 if it < 4 else 1.0)
        if it == 4:
            pr[:, :] = 1e-3
            pr[:, 0] = 1e3                                          # one dominant transition per env
        buf.update_priorities(s["step_choices"], pr)
    cap = buf.capacity
    sm, mn = buf.it_sum, buf.it_min
    assert torch.equal(sm[:, 1:cap], sm[:, 2:2 * cap:2] + sm[:, 3:2 * cap:2])
    assert torch.equal(mn[:, 1:cap], torch.minimum(mn[:, 2:2 * cap:2], mn[:, 3:2 * cap:2]))
    assert_close(sm[:, 1].cpu().numpy(), sm[:, cap:].sum(1).cpu().numpy(), 1e-12, "root")
    # sampling mass: the share of draws that land on a heavy leaf equals its share of the total priority mass
    dom = s["step_choices"][:, 0]
    u = torch.rand(n_envs, buf.per_env, dtype=torch.float64, device="cuda", generator=g)
    s2 = buf.sample(0.5, uniforms=u.cpu().numpy())
    share = (s2["step_choices"] == dom[:, None]).double().mean(1)
    leaves = sm[:, cap:cap + buf.size]
    expect = leaves.gather(1, dom[:, None])[:, 0] / sm[:, 1]
    assert float((share - expect).abs().max()) < 0.05 and float(expect.max()) > 0.1      # 64 stratified draws per env


def test_update_phase_is_deterministic_at_c2_size():
    """C2 (256 envs x 256 steps, 8 x 8 minibatches of 8 192): two agents built from the same seed produce bit-identical
    parameters after rollout + update phases (no atomics, fixed reduction orders, device-generated indices)."""
    from argparse import Namespace
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    out = []
    for _ in range(2):
        torch.manual_seed(0)
        cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128],
                        critic_hidden_size=[128], activation="leaky_relu", seed=1, parallels=256, running_steps=10 ** 7,
                        horizon_size=256, n_epochs=8, n_minibatch=8, learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01,
                        clip_range=0.2, gamma=0.98, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
                        grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
                        distributed_training=False, device="cuda", model_dir="/tmp/x", use_hip_graph=True)
        a = PPO_Agent(cfg, DeviceCartPoleVecEnv(256, seed=1))
        a.train(3 * 256)
        torch.cuda.synchronize()
        out.append((a.model.params.flat.clone(), a.memory.soa.fields["advantages"].clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_synthetic_marl_provider_semantics():
    """The SMAC-shaped provider (one launch per vector step): auto-reset, availability of action 0, episode limits,
    determinism per seed, reward formula."""
    from xuance_amd.envs import SyntheticSMACVecEnv
    n, T = 96, 12
    runs = []
    for rep in range(2):
        env = SyntheticSMACVecEnv(n, seed=5, max_episode_steps=T, p_term=0.05)
        env.reset()
        assert float(env.buf_avail[:, :, 0].min()) == 1.0 and int(env.steps.abs().sum()) == 0
        g = torch.Generator(device="cpu").manual_seed(0)
        trace, steps_host, totals = [], np.zeros(n, np.int64), np.zeros(2, np.int64)
        for t in range(40):
            state0 = env.buf_state
            acted = (env.buf_obs.clone(), env.buf_state.clone(), env.buf_avail.clone(), env.buf_obs, env.buf_avail)
            env.action.copy_(torch.randint(0, env.n_actions, (n, env.n_agents), generator=g, dtype=torch.int32))
            env.step_device()
            torch.cuda.synchronize()
            assert np.array_equal(env.prev_steps.cpu().numpy(), steps_host)
            assert env.buf_state.data_ptr() != state0.data_ptr()          # alternating buffers: the acted-on set is intact
            steps_host += 1
            term, trunc, done = env.terminated.cpu().numpy(), env.truncated.cpu().numpy(), env.done.cpu().numpy()
            totals += np.array([int((done > 0).sum()), int(steps_host[done > 0].sum())])
            assert np.array_equal(env.episode_totals.cpu().numpy(), totals)
            assert np.array_equal(done, np.maximum(term, trunc)) and not np.any((term > 0) & (trunc > 0))
            assert np.array_equal(trunc > 0, (term == 0) & (steps_host >= T))
            assert np.array_equal(env.end_step.cpu().numpy(), steps_host)
            steps_host[done > 0] = 0
            assert np.array_equal(env.steps.cpu().numpy(), steps_host)
            cont = torch.from_numpy(done == 0).cuda()
            assert torch.equal(env.buf_obs[cont], env.next_obs[cont]) and torch.equal(env.buf_state[cont], env.next_state[cont])
            if (~cont).any():
                assert not torch.equal(env.buf_obs[~cont], env.next_obs[~cont])
            assert float(env.buf_avail[:, :, 0].min()) == 1.0 and float(env.next_avail[:, :, 0].min()) == 1.0
            assert torch.equal(acted[0], acted[3]) and torch.equal(acted[1], state0) and torch.equal(acted[2], acted[4])
            rew = env.action.float().mean(1) / env.n_actions + 0.1 * state0[:, 0]
            assert torch.allclose(env.rewards, rew[:, None].expand(-1, env.n_agents), atol=1e-6)
            assert torch.equal(env.terminals, env.terminated[:, None].expand(-1, env.n_agents))
            trace.append(torch.cat([env.next_obs.flatten(), env.next_state.flatten(), env.next_avail.flatten(), env.done]).cpu())
        runs.append(torch.stack(trace))
    assert torch.equal(runs[0], runs[1])
    x = runs[0][:, :n * 3 * 30]
    assert abs(float(x.mean())) < 0.02 and abs(float(x.std()) - 1.0) < 0.02
    avail = runs[0][:, n * 3 * 30 + n * 48: n * 3 * 30 + n * 48 + n * 27].reshape(-1, 9)[:, 1:]
    assert abs(float(avail.mean()) - 0.7) < 0.02


def test_synthetic_frame_provider_semantics():
    """The Atari-shaped provider: alternating observation buffers (the acted-on frames survive the step), auto-reset,
    reward rule, episode limits, determinism per seed, uniform bytes."""
    from xuance_amd.envs import SyntheticAtariVecEnv
    n, T = 48, 9
    runs = []
    for rep in range(2):
        env = SyntheticAtariVecEnv(n, seed=4, max_episode_steps=T, p_term=0.05)
        env.reset()
        g = torch.Generator(device="cpu").manual_seed(0)
        steps_host, sums = np.zeros(n, np.int64), []
        for t in range(30):
            acted_on = env.buf_obs
            snapshot = acted_on.clone()
            env.action.copy_(torch.randint(0, 4, (n,), generator=g, dtype=torch.int32))
            env.step_device()
            torch.cuda.synchronize()
            assert env.buf_obs.data_ptr() != acted_on.data_ptr() and torch.equal(acted_on, snapshot)
            assert np.array_equal(env.reward.cpu().numpy(), (env.action.cpu().numpy() == steps_host % 4).astype(np.float32))
            steps_host += 1
            term, trunc, done = env.terminated.cpu().numpy(), env.truncated.cpu().numpy(), env.done.cpu().numpy()
            assert np.array_equal(done, np.maximum(term, trunc)) and np.array_equal(trunc > 0, (term == 0) & (steps_host >= T))
            assert np.array_equal(env.end_step.cpu().numpy(), steps_host)
            steps_host[done > 0] = 0
            assert np.array_equal(env.steps.cpu().numpy(), steps_host)
            cont = torch.from_numpy(done == 0).cuda()
            assert torch.equal(env.buf_obs[cont], env.next_obs[cont])
            if (~cont).any():
                assert not torch.equal(env.buf_obs[~cont], env.next_obs[~cont])
            sums.append(torch.cat([env.next_obs.flatten()[::97].float(), env.done]).cpu())
        runs.append(torch.stack(sums))
        x = env.next_obs.float()
        assert abs(float(x.mean()) - 127.5) < 0.5 and abs(float(x.std()) - 73.9) < 0.5
    assert torch.equal(runs[0], runs[1])
