"""GPU: end-to-end parity of the device rollout + update loop (PPO_Agent on DeviceCartPoleVecEnv) against the
oracle replaying the SAME trajectory: the device's sampled actions, reset observations and minibatch indices are
the fixed inputs (RNG streams cannot be matched), everything else -- physics, running
statistics, normalisation, values, log-probs, reward processing, path flags, GAE, advantages normalisation and all
minibatch updates -- is recomputed by the oracle and compared."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import assert_close, ChainCheck, _record

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_config(n_envs, T, **kw):
    c = dict(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128],
             critic_hidden_size=[128], activation="leaky_relu", seed=1, parallels=n_envs, running_steps=10 ** 6,
             horizon_size=T, n_epochs=2, n_minibatch=2, learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01,
             clip_range=0.2, gamma=0.98, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
             grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
             distributed_training=False, device="cuda", model_dir="/tmp/xrl_models", use_hip_graph=False)
    c.update(kw)
    return Namespace(**c)


def npy(t):
    return t.detach().cpu().numpy().copy()


@pytest.mark.parametrize("use_gae", [True, False])
def test_rollout_and_update_vs_oracle(oracle, use_gae):
    """use_gae=False: discounted-sum returns (memory_tools.py:258-261), where a terminated env must close its path with 0
    although bootv holds V(next_obs) -- bit 2 of `seg`."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    n, T = 24, 40
    torch.manual_seed(0)
    env = DeviceCartPoleVecEnv(n, seed=3)
    env.max_episode_steps = 25                     # force truncations inside the rollout
    # (use_post_norm False: this test looks at the buffer after EVERY vector step; with the default a step's bookkeeping rides in the
    #  next step's first launch -- test_device_act_tail_in_one_launch_equals_the_launches shows that form equal to this one)
    agent = PPO_Agent(make_config(n, T, use_gae=use_gae, use_post_norm=False), env)
    sd0 = {k: npy(v) for k, v in agent.model.state_dict().items()}
    env.reset()
    agent._started = True
    st = oracle.CartPoleOracle(npy(env.state))
    st.max_steps = 25
    obs_rms, ret_rms = oracle.RunningMeanStdOracle((4,)), oracle.RunningMeanStdOracle(())
    returns = np.zeros(n, np.float32)
    buf = oracle.OnPolicyBufferOracle((4,), (), n, T, gamma=0.98, gae_lam=0.95, use_gae=use_gae)
    raw_obs = npy(env.buf_obs)
    f = agent.memory.soa.fields
    for t in range(T):
        agent._enqueue_step(t)
        torch.cuda.synchronize()
        # ---- oracle mirror of ppo_agent.py:113-177 -------------------------------------------------------
        obs_rms.update(raw_obs)
        obs_n = oracle.process_observation(raw_obs, obs_rms).astype(np.float32)
        assert_close(npy(f["observations"][t]), obs_n, 1e-5, f"normalised obs t={t}")
        logits, value = oracle.actor_critic_forward(sd0, obs_n)
        acts = npy(f["actions"][t])                                   # fixed input: the device's sampled actions
        logp = oracle.log_softmax(logits)[np.arange(n), acts.astype(int)]
        assert_close(npy(f["values"][t]), value, 1e-5, "values")
        assert_close(npy(f["aux_old_logp"][t]), logp, 1e-5, "old_logp")
        next_obs, rew, term, trunc = st.step(acts.astype(int))
        assert_close(npy(env.next_obs), next_obs, 1e-6, "physics")
        assert np.array_equal(npy(env.terminated) > 0, term) and np.array_equal(npy(env.truncated) > 0, trunc)
        rew_n = oracle.process_reward(rew, ret_rms).astype(np.float32)
        assert_close(npy(f["rewards"][t]), rew_n, 1e-5, "normalised reward")
        buf.store(obs_n, acts, rew_n, value, term, {"old_logp": logp})
        vals_next, _ = None, None
        boot = oracle.actor_critic_forward(sd0, oracle.process_observation(next_obs, obs_rms).astype(np.float32))[1]
        if buf.full:                                                  # ppo_agent.py:129-135
            for i in range(n):
                buf.finish_path(0.0 if term[i] else boot[i], i)
            adv_full, ret_full = buf.advantages.copy(), buf.returns.copy()
        returns = (0.98 * returns + rew).astype(np.float32)
        done = term | trunc
        reset_obs = npy(env.buf_obs)
        for i in range(n):
            if done[i]:
                ret_rms.update(returns[i:i + 1])
                returns[i] = 0.0
                if not buf.full:
                    buf.finish_path(0.0 if term[i] else boot[i], i)
                st.state[i] = npy(env.state)[i]                       # fixed input: the device's reset state
                st.steps[i] = 0
        raw_obs = np.where(done[:, None], reset_obs, next_obs)
        assert_close(npy(agent.returns), returns, 1e-5, "return tracker")
        assert_close(npy(agent.ret_var)[0], ret_rms.var, 1e-5, "ret_rms.var")
        assert_close(npy(agent.obs_mean), obs_rms.mean, 1e-5, "obs_rms.mean")
        assert_close(npy(agent.obs_var), obs_rms.var, 1e-5, "obs_rms.var")
    assert int((npy(f["seg"]) & 1).sum()) > n                         # episodes really ended inside the rollout
    assert int(npy(env.truncated).sum()) >= 0
    # ---- close the rollout: bootstrap forward + GAE --------------------------------------------------------
    heads = agent.model.forward(agent.X, 2 * n)
    from xuance_amd import ops
    ops.policy_sample(heads=heads, act_out=None, val_out=None, logp_out=None, bootv_prev=f["bootv"][T - 1], n=n, A=2,
                      ld=3, gaussian=0, seed=1, step=0, step_dev=None)
    ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                 0.98, 0.95, use_gae)
    torch.cuda.synchronize()
    assert int(((npy(f["seg"]) & 4) > 0).sum()) == int((npy(f["terminals"]) > 0).sum()) > 0
    assert_close(npy(f["advantages"]).T, adv_full, 1e-5, "advantages", scale=float(np.abs(adv_full).max()))
    assert_close(npy(f["returns"]).T, ret_full, 1e-5, "returns", scale=float(np.abs(ret_full).max()))
    # ---- update phase ------------------------------------------------------------------------------------------
    agent.memory.ptr, agent.memory.size = 0, T
    info = agent.update()
    idx = npy(agent.idx)
    sd = {k: v.copy() for k, v in sd0.items()}
    opt = oracle.AdamOracle(sd, lr=4e-4, eps=1e-5, total_iters=agent.learner.total_iters)
    cfg = dict(vf_coef=0.25, ent_coef=0.01, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5)
    # the oracle buffer holds the oracle's own GAE; use it (not the device's) so the whole chain is independent
    for k in range(idx.shape[0]):
        s = buf.sample(idx[k])
        b = dict(obs=s["obs"], actions=s["actions"], returns=s["returns"], advantages=s["advantages"],
                 old_logp=s["aux_batch"]["old_logp"])
        oinfo, _ = oracle.ppo_update(sd, opt, b, cfg)
    for key in ("actor_loss", "critic_loss", "entropy", "predict_value"):
        ref = {"actor_loss": oinfo["a_loss"], "critic_loss": oinfo["c_loss"], "entropy": oinfo["e_loss"],
               "predict_value": oinfo["predict_value"]}[key]
        assert_close(info[key], ref, 1e-5, key)
    got = agent.model.state_dict()
    for k_, v in sd.items():
        assert_close(npy(got[k_]), v, 1e-5, f"param {k_} after {idx.shape[0]} updates")


def test_graph_replay_equals_eager():
    """The captured rollout/update graphs must reproduce the eager launch sequence bit for bit."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        env = DeviceCartPoleVecEnv(64, seed=5)
        agent = PPO_Agent(make_config(64, 32, use_hip_graph=use_graph, n_epochs=2, n_minibatch=4), env)
        idx = np.stack([np.random.default_rng(e).permutation(64 * 32) for e in range(2)]).reshape(8, -1)
        agent.set_indices(idx)
        infos = [agent.train(32) for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((npy(agent.model.params.flat), npy(agent.memory.soa.fields["advantages"]), infos[-1]))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2]["actor_loss"] == outs[1][2]["actor_loss"]


def test_replayed_rollout_graphs_are_deterministic():
    """Regression: the rollout graph (parameter re-pack -> scratch zeroing -> whole-rollout launch -> GAE) replayed after
    update phases must give the same bits in every repetition.  With a hipMemsetAsync node between the re-pack kernel and
    the whole-rollout launch, replays started the launch before the re-pack had finished (stale critic parameters in a
    workgroup now and then: 16 of 29 repetitions differed); the scratch is zeroed by a kernel now."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    ref = None
    for rep in range(8):
        torch.manual_seed(0)
        agent = PPO_Agent(make_config(64, 32, use_hip_graph=True, n_epochs=2, n_minibatch=4), DeviceCartPoleVecEnv(64, seed=3))
        snaps = []
        for it in range(3):
            agent.rollout()
            torch.cuda.synchronize()
            snaps.append({k: npy(v) for k, v in agent.memory.soa.fields.items()})
            agent.update()
        snaps.append({"params": npy(agent.model.params.flat)})
        assert agent.persist_status is not None
        if ref is None:
            ref = snaps
            continue
        for i, (a, b) in enumerate(zip(ref, snaps)):
            for k in a:
                assert np.array_equal(a[k], b[k]), f"repetition {rep}, snapshot {i}, field {k}"


def test_cartpole_learns():
    """Sanity (not parity): a few hundred thousand env steps of the fused loop must raise the episode score."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(1)
    env = DeviceCartPoleVecEnv(64, seed=1)
    agent = PPO_Agent(make_config(64, 256, use_hip_graph=True, n_epochs=8, n_minibatch=8, running_steps=64 * 256 * 30), env)
    agent.train(256 * 3)
    e0, s0, _ = env.episode_stats()
    env.stats.zero_()
    agent.train(256 * 27)
    e1, s1, _ = env.episode_stats()
    assert s0 < 60 and s1 > 100, (s0, s1)


@pytest.mark.parametrize("kernel", ["any-shape", "actor"])
def test_fused_rollout_step_equals_unfused_sequence(kernel):
    """xrl_rollout_step_cartpole (one launch per step, any shape) and xrl_rollout_cartpole_run + _values (the whole rollout of
    the 4-128-{128-2,128-1} class) must reproduce the seven-launch sequence per step they replace."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    res = []
    for fused in (False, True):
        torch.manual_seed(0)
        env = DeviceCartPoleVecEnv(100, seed=3)          # not a multiple of the 32-row / 16-row tiles
        env.max_episode_steps = 30
        agent = PPO_Agent(make_config(100, 48, use_fused_rollout=fused, use_actor_rollout=(kernel == "actor")), env)
        assert agent.use_fused_rollout == fused
        assert not fused or (agent._actor_rollout() is not None) == (kernel == "actor")
        agent.rollout()
        agent.rollout()                                   # second rollout: state carried across the boundary
        torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        if fused:
            stats = dict(obs_mean=npy(agent.pp["obs_stats"][0][:4]), obs_var=npy(agent.pp["obs_stats"][0][4:]),
                         ret_track=npy(agent.returns), eps=env.episode_stats())
        else:
            stats = dict(obs_mean=npy(agent.obs_mean), obs_var=npy(agent.obs_var), ret_track=npy(agent.returns),
                         eps=env.episode_stats())
        res.append((f, stats))
    (fa, sa), (fb, sb) = res
    assert np.array_equal(fa["actions"], fb["actions"]) and np.array_equal(fa["seg"], fb["seg"])
    assert np.array_equal(fa["terminals"], fb["terminals"])
    for k in ("observations", "values", "aux_old_logp", "rewards", "advantages", "returns"):
        assert_close(fb[k], fa[k], 2e-6, k, scale=max(1.0, float(np.abs(fa[k]).max())))
    need = fa["seg"] == 1                                 # bootstrap values are consumed where a path was cut without termination
    assert need.any()
    assert_close(fb["bootv"][need], fa["bootv"][need], 2e-6, "bootv", scale=max(1.0, float(np.abs(fa["bootv"][need]).max())))
    for k in ("obs_mean", "obs_var", "ret_track"):
        assert_close(sb[k], sa[k], 1e-6, k)
    assert sa["eps"][0] == sb["eps"][0] and sa["eps"][0] > 100


def _rollout_snapshot(agent, env):
    f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
    f.update(obs_stats=npy(agent.pp["obs_stats"][0]), ret_stats=npy(agent.pp["ret_stats"][0]),
             obs_count=npy(agent.pp["obs_count"][0]), ret_count=npy(agent.pp["ret_count"][0]),
             obs_raw=npy(agent.pp["obs_raw"][0]), ret_track=npy(agent.returns), cp_state=npy(env.state),
             cp_steps=npy(env.steps), cp_episodes=npy(env.episodes), eps=np.asarray(env.episode_stats()))
    return f


@pytest.mark.parametrize("act,n,norm", [("leaky_relu", 100, True), ("tanh", 64, True), ("relu", 37, True), ("sigmoid", 32, False),
                                        ("leaky_relu", 16, True), ("relu", 256, True)])
def test_actor_rollout_kernel_equals_any_shape_step_kernel(act, n, norm):
    """xrl_rollout_cartpole_run + xrl_rollout_cartpole_values (csrc/rollout_actor.hip: actor-only step chain on 16-row tiles,
    per-workgroup partial sums of the observation statistics, values as a batched pass afterwards) vs T launches of the
    any-shape step kernel (rollout_step_cartpole_kernel, 32-row tiles, values inside the step): the same trajectory --
    actions, episode ends and simulator state equal, everything real-valued to fp32 summation-order accuracy.  (An action
    could legitimately differ where the uniform lands within an ulp of the CDF; none does on these seeds.)"""
    from xuance_amd import ops
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    res = []
    for v2 in (False, True):
        torch.manual_seed(0)
        env = DeviceCartPoleVecEnv(n, seed=3)
        env.max_episode_steps = 30
        agent = PPO_Agent(make_config(n, 48, activation=act, use_obsnorm=norm, use_rewnorm=norm, use_actor_rollout=v2), env)
        assert agent.use_fused_rollout and (agent._actor_rollout() is not None) == v2
        agent.rollout()
        agent.rollout()
        torch.cuda.synchronize()
        if v2:
            assert agent.persist_status.tolist()[0] == 0
        res.append(_rollout_snapshot(agent, env))
    a, b = res
    assert a["eps"][0] > 50
    need = (a["seg"] == 1)                                          # bootstrap values exist where a path was cut without termination
    # (ret_stats / ret_count: the any-shape kernel merges a step's episode ends at the NEXT step, so its statistics lag by the
    # last step's ends between rollouts; their effect -- every step's normalised reward -- is compared)
    for k in ("actions", "terminals", "seg", "cp_steps", "cp_episodes", "obs_count"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("observations", "values", "aux_old_logp", "rewards", "advantages", "returns", "obs_stats", "obs_raw",
              "ret_track", "cp_state", "eps"):
        assert_close(b[k], a[k], 2e-6, k, scale=max(1.0, float(np.abs(a[k]).max())))
    assert need.any()
    assert_close(b["bootv"][need], a["bootv"][need], 2e-6, "bootv", scale=max(1.0, float(np.abs(a["bootv"][need]).max())))


@pytest.mark.parametrize("act,n,norm", [("leaky_relu", 100, True), ("tanh", 256, True), ("relu", 37, True), ("relu", 64, False), ("tanh", 16, True)])
def test_persistent_rollout_is_bit_identical_to_per_step_launches(act, n, norm):
    """actor_rollout_kernel as ONE launch per rollout (resident workgroups, one tagged message per workgroup and step) vs T
    launches of the same kernel with n_steps = 1 (state handed over in memory): every buffer field, statistic and simulator
    state must carry the same bits -- with the messages as plain stores in one L2 (what the launch geometry aims for;
    status[3] counts launches that did not get that placement) and forced through device-scope stores (the mode for any
    other placement)."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    res = []
    for persistent in (False, True, "coherent"):
        torch.manual_seed(0)
        env = DeviceCartPoleVecEnv(n, seed=3)
        env.max_episode_steps = 30
        agent = PPO_Agent(make_config(n, 48, activation=act, use_persistent_rollout=bool(persistent), use_obsnorm=norm,
                                      use_rewnorm=norm, persistent_coherent_exchange=(persistent == "coherent")), env)
        assert agent.use_fused_rollout and agent._actor_rollout() is not None
        agent.rollout()
        agent.rollout()
        torch.cuda.synchronize()
        if persistent:
            st = agent.persist_status.tolist()
            assert st[0] == 0, st                                      # no time-out
            exchanges = norm and n > 16                               # (one workgroup / no statistics: no messages at all)
            if persistent == "coherent":
                assert st[3] == (2 if exchanges else 0), st
            else:
                assert (st[3] == 0) == (bin(st[2]).count("1") <= 1), st
        else:
            assert getattr(agent, "persist_status", None) is None
        res.append(_rollout_snapshot(agent, env))
    a, b, c = res
    assert a["eps"][0] > 50
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]), k + " (device-scope exchange)"


@pytest.mark.parametrize("n,T,nmb", [(24, 40, 2), (64, 64, 4), (50, 30, 3)])
def test_fused_minibatch_kernel_equals_layered_path(n, T, nmb):
    """xrl_ppo_fused_minibatch (one launch) vs gather + grouped GEMMs + loss + backward GEMMs: same gradient and losses."""
    from xuance_amd import ops
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(0)
    env = DeviceCartPoleVecEnv(n, seed=2)
    agent = PPO_Agent(make_config(n, T, n_epochs=1, n_minibatch=nmb), env)
    agent.rollout()
    agent._new_indices()
    mem, lr = agent.memory, agent.learner
    assert lr.fused_eligible(mem)
    bs = agent.batch_size
    lr.prepare_buffer_update(mem, bs)
    lr.prepare_fused(mem, bs)
    lr.refresh_fused_params()
    ops.adv_stats(mem.soa.fields["advantages"], agent.idx.view(-1), bs, agent.idx.shape[0], n, T, lr.stats)
    k = agent.idx.shape[0] - 1
    lr.enqueue_minibatch_from_buffer(mem, agent.idx[k], lr.stats[k], finish=False)
    torch.cuda.synchronize()
    g_ref = npy(lr.optimizer.grad); info_ref = lr.last_info(bs); diag_ref = npy(lr.diag.view(-1)[:4 * bs])
    lr.optimizer.grad.zero_()
    lr.enqueue_minibatch_fused(mem, agent.idx[k], lr.stats[k], finish=False)
    torch.cuda.synchronize()
    g_fused = npy(lr.optimizer.grad); info_fused = lr.last_info(bs); diag_fused = npy(lr.diag.view(-1)[:4 * bs])
    scale = float(np.abs(g_ref).max())
    assert scale > 0
    assert_close(g_fused / scale, g_ref / scale, 1e-5, "gradient")
    for key in ("actor_loss", "critic_loss", "entropy", "predict_value", "clip_ratio"):
        # (the actor loss is a mean of surrogate terms that cancel almost exactly right after a rollout: scale = their magnitude)
        assert_close(info_fused[key], info_ref[key], 1e-5, key,
                     scale=float(np.abs(diag_ref[3 * bs:4 * bs]).mean()) if key == "actor_loss" else None)
    assert_close(diag_fused, diag_ref, 1e-5, "log_prob/ratio/surrogates")


@pytest.mark.parametrize("act,n,T,nmb", [("leaky_relu", 64, 64, 4), ("tanh", 50, 30, 3), ("relu", 256, 256, 8)])
def test_role_split_minibatch_kernel_vs_single_workgroup_kernel(act, n, T, nmb):
    """The role-split kernel on 32-row tiles (ppo_trunk_kernel<.., 32, 4, 2>: two workgroups per tile -- actor branch / critic branch --,
    two per CU) vs the any-shape single-workgroup kernel (ppo_fused_kernel, config.use_role_split_update: False) on the same
    minibatch: the per-sample diagnostics (log-prob, ratio, surrogates) and every branch / head gradient carry the SAME
    bits (same MFMA chains and reduction trees per element); the first-layer gradient -- whose two branch parts are now
    added by the slab reduction instead of in LDS -- and the loss sums agree to fp32 / fp64 re-association."""
    from xuance_amd import ops
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    res = []
    for split in (False, True):
        torch.manual_seed(0)
        agent = PPO_Agent(make_config(n, T, n_epochs=1, n_minibatch=nmb, activation=act, use_role_split_update=split, use_pair_update=False),
                          DeviceCartPoleVecEnv(n, seed=2))
        agent.rollout()
        agent._new_indices()
        mem, lr = agent.memory, agent.learner
        bs = agent.batch_size
        lr.prepare_buffer_update(mem, bs)
        lr.prepare_fused(mem, bs)
        assert lr.split == split
        lr.refresh_fused_params(mem, agent.idx)
        ops.adv_stats(mem.soa.fields["advantages"], agent.idx.view(-1), bs, agent.idx.shape[0], n, T, lr.stats)
        k = agent.idx.shape[0] - 1
        lr.enqueue_minibatch_fused(mem, agent.idx[k], lr.stats[k], finish=False)
        torch.cuda.synchronize()
        res.append(dict(grad=npy(lr.optimizer.grad), info=lr.last_info(bs), diag=npy(lr.diag.view(-1)[:4 * bs]),
                        off=dict(lr.model.params.offsets)))
    a, b = res
    assert np.array_equal(a["diag"], b["diag"])
    first = slice(0, 640)                                             # representation.model.0.{weight,bias}
    assert np.array_equal(a["grad"][640:], b["grad"][640:]), "branch / head gradients"
    scale = float(np.abs(a["grad"][first]).max())
    assert scale > 0 and np.abs(b["grad"][first]).max() > 0
    assert_close(b["grad"][first] / scale, a["grad"][first] / scale, 1e-6, "first-layer gradient")
    for key in a["info"]:
        assert_close(b["info"][key], a["info"][key], 1e-12 if key != "learning_rate" else 0, key)


def test_fused_reduce_adam_is_bit_identical_to_the_two_launches():
    """xrl_reduce_adam (slab reduction + clip + Adam + mirrors behind a counter barrier) vs xrl_grad_reduce followed by
    xrl_adam_step_mirrors: parameters, moments, clipped gradient and schedule state after whole update phases."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    res = []
    for fused_opt in (False, True):
        torch.manual_seed(0)
        agent = PPO_Agent(make_config(64, 32, n_epochs=2, n_minibatch=2, use_fused_optimizer=fused_opt),
                          DeviceCartPoleVecEnv(64, seed=5))
        for _ in range(2):
            agent.rollout()
            info = agent.update()
        torch.cuda.synchronize()
        lr, opt = agent.learner, agent.learner.optimizer
        st = opt.read()
        if fused_opt:
            assert lr.opt_sync.tolist()[:3] == [0, 0, 0]          # barrier reset itself, no time-out
        res.append(dict(p=npy(agent.model.params.flat), m=npy(opt.m), v=npy(opt.v), g=npy(opt.grad), frag=npy(lr.frag),
                        img=npy(lr.cache_image) if lr.cache_image is not None else np.zeros(1), sched=np.array([st.step, st.sched_steps, st.last_lr, st.last_grad_norm]),
                        info=np.array([info[k] for k in sorted(info)])))
    a, b = res
    assert a["sched"][0] == 8
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("split", [True, False])
def test_adam_mirrors_keep_every_derived_layout_current(split):
    """After full update phases the derived parameter copies equal a fresh re-pack of the parameters: the fragment-ordered copy of the
    branch layer (both sections) always; the transposed / packed copies of the any-shape minibatch kernel when that kernel is the
    one in use (role split declined) -- the role-split kernel reads neither, so its learner does not maintain them."""
    from xuance_amd import ops
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(0)
    agent = PPO_Agent(make_config(64, 32, n_epochs=2, n_minibatch=2, use_role_split_update=split), DeviceCartPoleVecEnv(64, seed=5))
    for _ in range(2):
        agent.rollout()
        agent.update()
    lr, plan, flat = agent.learner, agent.model.plan, agent.model.params.flat
    assert lr.split == split and lr.frag is not None and len(lr._mirrors) == (2 if split else 4)
    fr = torch.zeros_like(lr.frag)
    ops.pack_mid_frags(plan, flat, fr)
    torch.cuda.synchronize()
    assert torch.equal(lr.frag, fr)
    if not split:
        pt, img = torch.zeros_like(lr.params_t), torch.zeros_like(lr.cache_image)
        ops.transpose_mid(plan, flat, pt); ops.pack_rollout_cache(plan, flat, img)
        torch.cuda.synchronize()
        mid = [L for st in plan.stages[1:-1] for L in st]
        lo = agent.model.params.offsets[mid[0].w_name]; hi = lo + mid[0].N * mid[0].K
        assert torch.equal(lr.params_t[lo:hi], pt[lo:hi]) and torch.equal(lr.cache_image, img)


@pytest.mark.parametrize("wide,use_graph,n,T,nmb,whole", [(True, False, 32, 16, 2, False), (True, True, 32, 16, 2, False),
                                                          (False, False, 32, 16, 2, False), (True, True, 128, 256, 8, False),
                                                          (True, False, 32, 16, 2, True), (True, True, 128, 256, 8, True),
                                                          (True, True, 32, 16, 2, True)])
def test_ppo_gaussian_agent_on_mujoco_shape(oracle, wide, use_graph, n, T, nmb, whole):
    """C4 shapes (obs 17, Box(6), Gaussian actor 17-256-256-6 tanh, critic 17-256-256-1, Basic_Identical): rollout + update
    end to end, checked against the oracle on the device's own rollout data.  wide: the update runs as ONE launch per
    minibatch (xrl_ppo_wide_minibatch: csrc/ppo_wide.hip) from rows gathered once per phase; otherwise the layered path.
    (128, 256, 8): the BASELINE configs[3] size per GPU -- 128 envs x horizon 256, minibatches of 4 096, graphs on, acting launch
    with the running statistics and the bookkeeping inside; ONE epoch of 8 chained minibatches instead of the config's 16 x 8, so
    that the oracle's replay of the update chain stays a few seconds.  whole: the rollout as ONE launch with only the actor on the
    step chain + batched values (xrl_rollout_wide_run, csrc/rollout_wide.hip: what tools/bench_c4.py times since round 4) instead
    of the launches per vector step; every check against the oracle is the same, and the two forms are compared with each other
    in test_wide_rollout_launch_matches_the_launches_per_step."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticMujocoVecEnv
    torch.manual_seed(0)
    ms = 10 if T == 16 else 100                                        # episode cut-off of the provider (truncations inside the rollout)
    cfg = make_config(n, T, representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                      critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", n_epochs=1,
                      n_minibatch=nmb, ent_coef=0.0, gamma=0.99, use_hip_graph=use_graph, use_fused_update=wide, use_wide_rollout=whole)
    env = SyntheticMujocoVecEnv(n, seed=4, max_episode_steps=ms)
    agent = PPO_Agent(cfg, env)
    assert agent.model.dist == "gaussian" and not agent.use_fused_rollout and (agent._wide_rollout() is not None) == whole
    assert agent.learner.wide_eligible() == wide and agent.learner.fused_eligible(agent.memory) == wide
    assert sum(int(np.prod(v.shape)) for v in agent.model.state_dict().values()) == 142605    # SURVEY 8a parameter count
    sd = {k: npy(v) for k, v in agent.model.state_dict().items()}
    agent.rollout()
    torch.cuda.synchronize()
    f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
    # acting: values and log-probs of the stored (observation, action) pairs under the initial parameters
    mu, v = oracle.actor_critic_forward(sd, f["observations"].reshape(-1, 17), "gaussian", "leaky_relu", "tanh")
    assert_close(f["values"].reshape(-1), v, 1e-5, "values")
    ls = sd["actor.log_std"]
    x = f["actions"].reshape(-1, 6)
    lp = (-((x - mu) ** 2) / (2 * np.exp(ls) ** 2) - ls - 0.5 * np.log(2 * np.pi)).sum(-1)
    assert_close(f["aux_old_logp"].reshape(-1), lp, 1e-5, "old_logp", scale=float(np.abs(lp).max()))
    # the actions themselves: mu + std * N(0, 1) with the engine's Philox draws (first rollout: global step = t)
    mu3 = mu.reshape(T, n, 6)
    for t_ in (0, 1, T - 1):
        z = oracle.action_gaussians(agent.seed, n, t_, 6)
        assert_close(f["actions"][t_], mu3[t_] + np.exp(ls) * z, 1e-5, f"actions at step {t_}", scale=4.0)
    if ms - 1 < T:
        assert (f["seg"][ms - 1] & 1).all()                                # truncation at the cut-off ...
    assert (f["seg"][T - 1] & 1).all()                                     # ... and at buffer end
    if whole:
        assert agent._wr_status.tolist()[0] == 0
    # update: oracle on the same minibatches
    idx = np.stack([np.random.default_rng(3).permutation(n * T)]).reshape(nmb, -1)
    agent.set_indices(idx)
    info = agent.update()
    buf = oracle.OnPolicyBufferOracle((17,), (6,), n, T)
    buf.size = T
    buf.observations, buf.actions = f["observations"].transpose(1, 0, 2), f["actions"].transpose(1, 0, 2)
    buf.returns, buf.values, buf.advantages, buf.old_logp = f["returns"].T, f["values"].T, f["advantages"].T, f["aux_old_logp"].T
    opt = oracle.AdamOracle(sd, lr=4e-4, eps=1e-5, total_iters=agent.learner.total_iters)
    c = dict(vf_coef=0.25, ent_coef=0.0, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5)
    sd0 = {k: v.copy() for k, v in sd.items()}
    chain = ChainCheck(4e-4, total_iters=agent.learner.total_iters)
    long_chain = nmb > 2
    sd64 = {k: v.astype(np.float64) for k, v in sd.items()}
    opt64 = oracle.AdamOracle(sd64, lr=4e-4, eps=1e-5, total_iters=agent.learner.total_iters)
    for k in range(nmb):
        s = buf.sample(idx[k])
        b = dict(obs=s["obs"], actions=s["actions"], returns=s["returns"], advantages=s["advantages"], old_logp=s["aux_batch"]["old_logp"])
        oi, _ = oracle.ppo_update(sd, opt, b, c, dist="gaussian", act="leaky_relu", activation_action="tanh")
        chain.step(oi["clipped_grads"])
        if long_chain:                                                 # the same chain in float64 (same float32 inputs)
            oracle.ppo_update(sd64, opt64, {k_: np.asarray(v_, np.float64) for k_, v_ in b.items()}, c,
                              dist="gaussian", act="leaky_relu", activation_action="tanh")
    got = {k_: npy(v_) for k_, v_ in agent.model.state_dict().items()}
    if not long_chain:
        # the distance each tensor moved in two updates, held to what gradients agreeing at 1e-5 of their scale allow
        chain.check(got, sd, sd0)
    else:
        # Eight chained 4 096-row updates: PPO's clipped surrogate has a discontinuous gradient at ratio = 1 +- eps -- one sample of a
        # minibatch within float32 rounding of that boundary contributes its whole gradient in one float32 evaluation and nothing in
        # another (2.4e-4 of the minibatch's actor gradient), which no propagated rounding bound covers (measured round 4: the
        # propagated bound held on one provider's data and was missed by 1e-3 of the distance moved on another's, launches per step
        # and whole-rollout launch alike).  As in test_gpu_headline.py the yardstick is the float32 ORACLE's own distance from the
        # float64 chain on the same minibatches.
        DEV_K = 32.0
        for k_, x64 in sd64.items():
            S = float(np.abs(x64 - sd0[k_]).max()) or 1.0              # the distance the tensor moved
            dev, ref = float(np.abs(got[k_] - x64).max()) / S, float(np.abs(sd[k_] - x64).max()) / S
            _record(f"C4 chain {k_} after {nmb} updates: engine vs f64 chain [f32 oracle vs f64: {ref:.3e}]", dev, ref, 0.0, x64.size)
            assert dev <= max(1e-5, DEV_K * ref), f"param {k_} after {nmb} updates: engine {dev:.3e}, float32 oracle {ref:.3e} of the distance moved"
    assert_close(info["critic_loss"], oi["c_loss"], 1e-5, "critic_loss")
    assert_close(info["actor_loss"], oi["a_loss"], 1e-5, "actor_loss", scale=float(np.abs(oi["surrogate2"]).mean()))
    if wide and not whole:
        # running statistics + normalisation inside the acting launch == xrl_obs_normalize as a launch of its own, bit for bit
        torch.manual_seed(0)
        cfg2 = make_config(n, T, representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                           critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", n_epochs=1,
                           n_minibatch=nmb, ent_coef=0.0, gamma=0.99, use_hip_graph=False, use_fused_obsnorm=False, use_wide_rollout=False)
        b = PPO_Agent(cfg2, SyntheticMujocoVecEnv(n, seed=4, max_episode_steps=ms))
        assert agent._wstats is not None and agent._wpost and b._wide_acting() is not None and b._wstats is None and not b._wpost
        b.rollout()
        torch.cuda.synchronize()
        f2 = {k: npy(v) for k, v in b.memory.soa.fields.items()}
        for k in ("observations", "actions", "values", "aux_old_logp", "rewards", "terminals", "seg", "advantages", "returns", "bootv"):
            assert np.array_equal(f[k], f2[k]), k              # (incl. the bookkeeping that rode in the acting launches)
        for x, y in ((agent.ret_mean, b.ret_mean), (agent.ret_var, b.ret_var), (agent.ret_count, b.ret_count), (agent.returns, b.returns)):
            assert torch.equal(x, y)
    if wide and use_graph:
        # run-to-run determinism of the whole path (the acting launch's four-workgroup hand-off sums in part order whoever
        # arrives last; gradient rows are reduced in row order): a second agent from the same seeds lands on the same bits
        torch.manual_seed(0)
        c = PPO_Agent(make_config(n, T, representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                                  critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", n_epochs=1,
                                  n_minibatch=nmb, ent_coef=0.0, gamma=0.99, use_hip_graph=True, use_fused_update=True, use_wide_rollout=whole),
                      SyntheticMujocoVecEnv(n, seed=4, max_episode_steps=ms))
        c.rollout()
        c.set_indices(idx)
        c.update()
        torch.cuda.synchronize()
        for k, v in c.memory.soa.fields.items():
            assert np.array_equal(npy(v), f[k]), k
        assert torch.equal(c.model.params.flat, agent.model.params.flat)
    if wide:                            # the optimiser launch kept the fragment-ordered copy of the middle layers current
        lr = agent.learner
        fr = lr._wide.frag.clone()
        lr._wide.pack()
        torch.cuda.synchronize()
        assert torch.equal(fr, lr._wide.frag)


@pytest.mark.parametrize("n,T,norm,ms", [(128, 48, True, 20), (100, 32, True, 10), (16, 32, True, 12), (64, 32, False, 10)])
def test_wide_rollout_launch_matches_the_launches_per_step(n, T, norm, ms):
    """xrl_rollout_wide_run (ONE launch per rollout: actor-only step chain on 16-row tiles, per-workgroup partial sums of the
    observation statistics, dynamics and records inside, batched values afterwards) vs the launches per vector step
    (xrl_wide_act_step + xrl_synth_control_step + bookkeeping): same Philox draws, same dynamics arithmetic -- two rollouts agree to
    fp32 summation-order accuracy on every field (continuous actions: nothing can flip), episode ends equal; and the whole-rollout
    launch equals the same kernel launched once per vector step bit for bit."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticMujocoVecEnv
    res = []
    for mode in ("steps", "whole", "whole-per-step"):
        torch.manual_seed(0)
        cb = None
        cfg = make_config(n, T, representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                          critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", n_epochs=1, n_minibatch=2,
                          ent_coef=0.0, gamma=0.99, use_hip_graph=mode != "whole-per-step", use_obsnorm=norm, use_rewnorm=norm,
                          use_wide_rollout=mode != "steps")
        env = SyntheticMujocoVecEnv(n, seed=4, max_episode_steps=ms)
        agent = PPO_Agent(cfg, env)
        assert (agent._wide_rollout() is not None) == (mode != "steps")
        if mode == "whole-per-step":
            agent._per_step = lambda: True                             # (what a per-step callback makes of the rollout)
        agent.rollout()
        agent.rollout()
        torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        f.update(obs_mean=npy(agent.obs_mean), obs_var=npy(agent.obs_var), obs_count=npy(agent.obs_count), ret_mean=npy(agent.ret_mean),
                 ret_var=npy(agent.ret_var), ret_count=npy(agent.ret_count), ret_track=npy(agent.returns), state=npy(env.state),
                 steps=npy(env.steps), stats=npy(env.stats), buf_obs=npy(env.buf_obs))
        res.append(f)
    a, b, c = res
    assert a["stats"][0] > 0                                           # episodes ended inside the rollouts
    for k in ("seg", "terminals", "steps", "obs_count", "ret_count"):
        assert np.array_equal(a[k], b[k]), k
    need = a["seg"] == 1
    for k in ("observations", "actions", "values", "aux_old_logp", "rewards", "advantages", "returns", "obs_mean", "obs_var", "ret_mean",
              "ret_var", "ret_track", "state", "buf_obs", "stats"):
        assert_close(b[k], a[k], 2e-5, k, scale=max(1.0, float(np.abs(a[k]).max())))
    assert_close(b["bootv"][need], a["bootv"][need], 2e-5, "bootv", scale=max(1.0, float(np.abs(a["bootv"][need]).max())))
    for k in b:
        assert np.array_equal(b[k], c[k]), k + " (one launch per rollout vs the same kernel once per vector step)"


@pytest.mark.parametrize("n,T,D,A,act,oact,obsnorm", [(48, 16, 17, 6, "leaky_relu", "tanh", True),     # tiles straddle row n
                                                      (32, 15, 17, 6, "leaky_relu", "tanh", True),     # odd horizon
                                                      (64, 8, 24, 8, "relu", None, True),              # largest D, A
                                                      (32, 8, 3, 1, "tanh", "tanh", False),            # smallest, no obs norm
                                                      (96, 8, 17, 6, "relu", "tanh", True)])
def test_wide_acting_launch_matches_the_layered_rollout(n, T, D, A, act, oact, obsnorm):
    """xrl_wide_act_step in every configuration the agent can put it in (with / without the statistics and the bookkeeping
    inside the launch -- env counts that are not tile multiples and odd horizons fall back to launches of their own) against
    the layered rollout (use_fused_acting: False: xrl_obs_normalize + three GEMM launches + xrl_policy_sample +
    xrl_rollout_poststep) from the same seeds: statistics and bookkeeping bit for bit, network outputs within 1e-5 (the
    dot products are summed in another order), over a whole rollout with episode ends, followed by one update phase."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticMujocoVecEnv
    out = {}
    for fused in (True, False):
        torch.manual_seed(0)
        cfg = make_config(n, T, representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                          critic_hidden_size=[256, 256], activation=act, activation_action=oact, n_epochs=1, n_minibatch=2,
                          ent_coef=0.0, gamma=0.99, use_hip_graph=fused, use_fused_acting=fused, use_obsnorm=obsnorm,
                          use_wide_rollout=False)                                 # (the launches per vector step are what is tested here)
        agent = PPO_Agent(cfg, SyntheticMujocoVecEnv(n, seed=4, obs_dim=D, act_dim=A, max_episode_steps=5))
        agent.rollout()
        torch.cuda.synchronize()
        assert (agent._wide_acting() is not None) == fused
        if fused:
            assert (agent._wstats is not None) == (T % 2 == 0 and n % 32 == 0)
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        st = dict(obs_mean=npy(agent.obs_mean), obs_var=npy(agent.obs_var), obs_count=npy(agent.obs_count),
                  ret_mean=npy(agent.ret_mean), ret_var=npy(agent.ret_var), ret_count=npy(agent.ret_count))
        info = agent.update()
        out[fused] = (f, st, info, npy(agent.model.params.flat))
    (f1, s1, i1, p1), (f0, s0, i0, p0) = out[True], out[False]
    # the first vector step sees identical inputs: statistics and normalised observations agree bit for bit there; later
    # steps act on actions that differ in the last bits, so trajectories are compared with the tolerance of the whole path
    assert np.array_equal(f1["observations"][0], f0["observations"][0])
    assert_close(f1["values"][0], f0["values"][0], 1e-5, "values[0]")
    assert_close(f1["actions"][0], f0["actions"][0], 1e-5, "actions[0]", scale=4.0)
    assert_close(f1["aux_old_logp"][0], f0["aux_old_logp"][0], 1e-5, "logp[0]", scale=float(np.abs(f0["aux_old_logp"][0]).max()))
    assert np.array_equal(f1["terminals"], f0["terminals"]) and np.array_equal(f1["seg"], f0["seg"])
    for k in ("observations", "actions", "values", "rewards", "advantages", "returns"):
        assert_close(f1[k], f0[k], 2e-4, k, scale=max(1.0, float(np.abs(f0[k]).max())))
    for k in s0:
        assert_close(s1[k], s0[k], 1e-5, k)
    assert np.isfinite(p1).all() and np.isfinite(i1["actor_loss"])


@pytest.mark.parametrize("use_graph", [False, True])
def test_a2c_agent_vs_oracle(oracle, use_graph):
    """A2C_Agent (a2c_agent.py:18-79 + a2c_learner.py:34-90) on the device CartPole: ActorCritic model with one trunk
    per head, the rollout kernels shared with PPO, one whole-buffer update per rollout; the oracle replays the update
    on the device's own rollout data."""
    from xuance_amd.agents import A2C_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(0)
    n, T = 16, 16
    cfg = make_config(n, T, n_epochs=1, n_minibatch=1, running_steps=4000, end_factor_lr_decay=0.5, use_hip_graph=use_graph)
    agent = A2C_Agent(cfg, DeviceCartPoleVecEnv(n, seed=5))
    assert list(agent.model.plan.widths) == [4, 256, 256, 3] and agent.learner.loss_mode == 1
    # (state_dict() speaks the reference ActorCritic's key names -- actor.representation.model.*, actor.actor_head.logits.*, ... --;
    #  the oracle's layer chains are named like the engine's internal ones)
    assert list(agent.model.state_dict())[:3] == ["actor.representation.model.0.weight", "actor.representation.model.0.bias", "actor.actor_head.logits.0.weight"]
    sd = {n_: npy(agent.model.params.view(n_)) for n_ in agent.model.ref_order}
    opt = oracle.AdamOracle(sd, lr=4e-4, eps=1e-5, end_factor=0.5, total_iters=4000)
    c = dict(vf_coef=0.25, ent_coef=0.01, use_grad_clip=True, grad_clip_norm=0.5)
    for it in range(3 if use_graph else 2):                       # the third pass replays the captured graphs
        agent.rollout()
        torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        logits, v = oracle.actor_critic_forward(sd, f["observations"].reshape(-1, 4))
        assert_close(f["values"].reshape(-1), v, 1e-5, "values")
        idx = np.random.default_rng(it).permutation(n * T).reshape(1, -1)
        agent.set_indices(idx)
        info = agent.update()
        buf = oracle.OnPolicyBufferOracle((4,), (), n, T)
        buf.size = T
        buf.observations, buf.actions = f["observations"].transpose(1, 0, 2), f["actions"].T
        buf.returns, buf.values, buf.advantages, buf.old_logp = f["returns"].T, f["values"].T, f["advantages"].T, f["aux_old_logp"].T
        s = buf.sample(idx[0])
        oi, _ = oracle.ppo_update(sd, opt, dict(obs=s["obs"], actions=s["actions"], returns=s["returns"],
                                                advantages=s["advantages"]), c, loss_kind="a2c")
        for k_, val in sd.items():
            assert_close(npy(agent.model.params.view(k_)), val, 1e-5, f"param {k_} after update {it}")
        assert_close(info["actor-loss"], oi["a_loss"], 1e-5, "actor-loss")
        assert_close(info["critic-loss"], oi["c_loss"], 1e-5, "critic-loss")
        assert_close(info["learning_rate"], oi["learning_rate"], 1e-9, "lr")
    assert "clip_ratio" not in info


@pytest.mark.parametrize("fused", [True, False])
def test_agent_checkpoint_files(tmp_path, fused):
    """Agent.save_model / load_model (agent.py:199-230): `<name>.pth` in the learner layout + `obs_rms.npy` holding a dict
    {'count', 'mean', 'var'}; a fresh agent that loads them acts and normalises exactly like the saved one."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    n, T = 32, 16
    cfg = make_config(n, T, use_fused_rollout=fused, use_hip_graph=False, model_dir=str(tmp_path))
    a = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=3))
    a.train(2 * T)
    a.save_model("final_train_model.pth", model_path=str(tmp_path))        # (default: a seed_* run folder under model_dir)
    st = np.load(tmp_path / "obs_rms.npy", allow_pickle=True).item()
    assert set(st) == {"count", "mean", "var"} and st["mean"].shape == (4,) and st["mean"].dtype == np.float32
    assert st["count"] > 2 * T * n - 1
    ck = torch.load(tmp_path / "final_train_model.pth", weights_only=True)
    assert set(ck) == {"policy", "optimizer", "rng_state", "cuda_rng_state"}
    b = PPO_Agent(make_config(n, T, use_fused_rollout=fused, use_hip_graph=False, model_dir=str(tmp_path)),
                  DeviceCartPoleVecEnv(n, seed=3))
    b.load_model(str(tmp_path), "final_train_model.pth")
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[k]), k
    for x, y in zip(a._obs_stats_tensors(), b._obs_stats_tensors()):
        assert torch.equal(x.float(), y.float())
    assert a.learner.optimizer.read().step == b.learner.optimizer.read().step
    # same env seed, same RNG counters from construction: the restored agent's first rollout == a fresh agent's first
    # rollout evaluated with the restored parameters and statistics (values of step 0 depend on both)
    b.rollout(); torch.cuda.synchronize()
    obs0 = npy(b.memory.soa.fields["observations"][0])
    raw = npy(b.envs.buf_obs) if False else None
    assert np.isfinite(obs0).all() and np.abs(obs0).max() <= 5.0


def test_shm_subproc_vec_env_to_device():
    """ShmSubprocVecEnv.step_to_device: a vector step of host envs (worker processes) reaches HBM through ONE async copy
    of the page-locked shared block, and from there the on-policy buffer; device tensors == the host arrays."""
    from xuance_amd.envs import ShmSubprocVecEnv, NumpyCartPoleEnv
    from xuance_amd.memory import HipOnPolicyBuffer
    from xuance_amd.spaces import Discrete
    n, T = 8, 12
    venv = ShmSubprocVecEnv([NumpyCartPoleEnv] * n, env_seed=3, in_series=2, device="cuda")
    try:
        assert venv._pinned, "hipHostRegister of the shared block failed"
        buf = HipOnPolicyBuffer(venv.observation_space, Discrete(2), {"old_logp": ()}, n, T)
        obs, _ = venv.reset()
        rng = np.random.default_rng(1)
        host = []
        for t in range(T):
            acts = rng.integers(0, 2, n).astype(np.float32)
            d = venv.step_to_device(torch.from_numpy(acts).cuda())
            torch.cuda.synchronize()
            h = {k: venv.v[k].copy() for k in ("obs", "rewards", "terminated", "truncated")}
            for k in h:
                assert np.array_equal(d[k].cpu().numpy(), h[k]), k
            buf.store(d["obs"], torch.from_numpy(acts).cuda(), d["rewards"], torch.zeros(n, device="cuda"),
                      d["terminated"].float(), {"old_logp": torch.zeros(n, device="cuda")})
            host.append(h)
        f = buf.soa.fields
        for t in range(T):
            assert np.array_equal(npy(f["observations"][t]), host[t]["obs"])
            assert np.array_equal(npy(f["rewards"][t]), host[t]["rewards"])
    finally:
        venv.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_ragged_minibatches_vs_oracle(oracle, use_graph):
    """buffer_size not divisible by n_minibatch (5 envs x 10 steps, 4 minibatches -> 12, 12, 12, 12, 2 per epoch): the
    reference's train_epochs ends every epoch with a short minibatch (on_policy.py:198-203); the oracle replays the same
    permutations."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(0)
    n, T = 5, 10
    cfg = make_config(n, T, n_epochs=2, n_minibatch=4, use_hip_graph=use_graph, use_fused_rollout=False)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=2))
    assert agent.batch_size == 12 and agent.rem == 2
    sd = {k: npy(v) for k, v in agent.model.state_dict().items()}
    opt = oracle.AdamOracle(sd, lr=4e-4, eps=1e-5, total_iters=agent.learner.total_iters)
    c = dict(vf_coef=0.25, ent_coef=0.01, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5)
    for it in range(3 if use_graph else 1):
        agent.rollout()
        torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        perms = np.stack([np.random.default_rng(10 * it + e).permutation(n * T) for e in range(2)])
        agent.set_indices(perms)
        info = agent.update()
        buf = oracle.OnPolicyBufferOracle((4,), (), n, T)
        buf.size = T
        buf.observations, buf.actions = f["observations"].transpose(1, 0, 2), f["actions"].T
        buf.returns, buf.values, buf.advantages, buf.old_logp = f["returns"].T, f["values"].T, f["advantages"].T, f["aux_old_logp"].T
        n_updates = 0
        for e in range(2):
            for start in range(0, n * T, 12):
                s = buf.sample(perms[e, start:start + 12])
                oi, _ = oracle.ppo_update(sd, opt, dict(obs=s["obs"], actions=s["actions"], returns=s["returns"],
                                                        advantages=s["advantages"], old_logp=s["aux_batch"]["old_logp"]), c)
                n_updates += 1
        assert n_updates == 10 and len(s["obs"]) == 2
        got = agent.model.state_dict()
        for k_, val in sd.items():
            assert_close(npy(got[k_]), val, 1e-5, f"param {k_} (pass {it})")
        assert_close(info["critic_loss"], oi["c_loss"], 1e-5, "critic_loss of the short minibatch")
    assert agent.learner.iterations == 10 * (3 if use_graph else 1)


@pytest.mark.parametrize("use_graph", [False, True])
def test_pg_agent_rollout_and_update(use_graph):
    """PG_Agent (pg_agent.py:12-79): actor-only model, stored values 0, paths closed with the processed reward of the cut
    step, non-GAE returns; the whole-buffer update through the agent equals PG_Learner.update (pinned by
    tests/golden/pg_*.npz) on the same sampled batch."""
    from xuance_amd.agents import REGISTRY_Agents
    from xuance_amd.envs import DeviceCartPoleVecEnv
    from xuance_amd.learners import PG_Learner
    from xuance_amd.nets import ActorNet
    torch.manual_seed(0)
    n, T, gamma = 16, 24, 0.98
    cfg = make_config(n, T, n_epochs=1, n_minibatch=1, running_steps=4000, use_hip_graph=use_graph, use_gae=False,
                      use_advnorm=False, activation="relu", representation_hidden_size=[128], actor_hidden_size=[128])
    env = DeviceCartPoleVecEnv(n, seed=5)
    env.max_episode_steps = 9                                     # truncations inside the horizon: paths cut with a bootstrap
    agent = REGISTRY_Agents["PG"](cfg, env)
    assert isinstance(agent.model, ActorNet) and isinstance(agent.learner, PG_Learner) and not agent.use_fused_rollout
    for it in range(3 if use_graph else 2):
        agent.rollout()
        torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        # closing value of a cut path = the processed reward of its last step over the return statistics of THAT moment: equal to the
        # stored reward except for an env whose own episode end (and those of the envs before it) updated ret_rms first
        # (on_policy.py:272-283; pinned to the reference's run by tests/test_gpu_agent_replay.py::test_pg_agent_replays_the_reference_run)
        mid = ((f["seg"] & 1) > 0)
        mid[T - 1] = False
        assert not f["values"].any() and np.array_equal(f["bootv"][~mid], f["rewards"][~mid])
        assert (f["bootv"][mid] > 0).all() and (f["bootv"][mid] <= 5.0).all()        # (here both saturate at rewnorm_range: the return std is below its 0.1 floor)
        # returns: discounted sums inside a path; a path ends where seg is set, bootstrapped with the processed reward of that
        # step unless the env terminated there (on_policy.py:246-252, 263-268; memory_tools.py:259-263)
        ret = np.zeros((T, n), np.float64)
        for e in range(n):
            nxt = 0.0
            for t in reversed(range(T)):
                if f["seg"][t, e] & 1:
                    assert bool(f["seg"][t, e] & 4) == bool(f["terminals"][t, e])       # terminated: finish_path(0.0, i)
                    nxt = 0.0 if f["terminals"][t, e] else float(f["bootv"][t, e])
                ret[t, e] = f["rewards"][t, e] + gamma * nxt
                nxt = ret[t, e]
        assert (f["seg"][T - 1] & 1).all() and (f["seg"][:T - 1] & 1).any()
        assert_close(f["returns"], ret.astype(np.float32), 2e-6, "returns")
        # advantages of the non-GAE branch: rewards[:-1] + gamma * vs[1:] - vs[:-1] with vs = 0 inside a path and `val` at
        # its end (memory_tools.py:261) -- PG_Learner does not read them
        end = (f["seg"] & 1) > 0
        val = np.where(f["terminals"] > 0, 0.0, f["bootv"])
        assert_close(f["advantages"], (f["rewards"] + np.where(end, gamma * val, 0.0)).astype(np.float32), 2e-6, "advantages")
        # the agent's update against a stand-alone PG_Learner on the same batch, from the same parameters
        twin = ActorNet(4, 2, "categorical", (128,), (128,), "relu")
        twin.load_state_dict(agent.model.state_dict())
        tl = PG_Learner(cfg, twin, None)
        tl.optimizer.load_state_dict(agent.learner.optimizer.state_dict())
        tl.iterations = agent.learner.iterations
        idx = np.random.default_rng(it).permutation(n * T).reshape(1, -1)
        agent.set_indices(idx)
        info = agent.update()
        env_i, t_i = np.divmod(idx[0], T)
        ti = tl.update(obs=f["observations"][t_i, env_i], actions=f["actions"][t_i, env_i], returns=f["returns"][t_i, env_i],
                       batch_size=n * T)
        assert set(info) == set(ti) == {"actor-loss", "entropy", "learning_rate"}
        for k in ti:
            assert_close(info[k], ti[k], 1e-6, k)
        got, ref = agent.model.state_dict(), twin.state_dict()
        for k in ref:
            assert_close(npy(got[k]), npy(ref[k]), 1e-6, f"param {k} after update {it}")


def test_unusable_whole_rollout_launch_falls_back_and_redoes_the_rollout():
    """The whole-rollout launch reports a barrier time-out in status[0] (results invalid).  Fault injected by a memset of
    that word captured in front of the launch: the agent must notice on its FIRST rollout (read synchronously), restore
    simulator / statistics / counters, fall back to per-step launches for good and redo the rollout -- ending
    bit-identical to an agent that never used the whole-rollout launch."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    res = []
    for inject in (False, True):
        torch.manual_seed(0)
        env = DeviceCartPoleVecEnv(64, seed=3)
        env.max_episode_steps = 30
        agent = PPO_Agent(make_config(64, 48, use_hip_graph=True, use_persistent_rollout=inject), env)
        if inject:
            orig = agent._persistent_ok

            def faulty():
                ok = orig()
                if ok:
                    agent.persist_status[0:1].fill_(1)                # "a barrier timed out"
                return ok
            agent._persistent_ok = faulty
            with pytest.warns(UserWarning, match="whole-rollout launch unusable"):
                agent.rollout()
            assert agent.persist_status is None and agent.config.use_persistent_rollout is False
        else:
            agent.rollout()
        agent.rollout()
        info = agent.update()                                         # no status to complain about
        torch.cuda.synchronize()
        f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
        f.update(cp_state=npy(env.state), cp_episodes=npy(env.episodes), ret_track=npy(agent.returns),
                 obs_stats=npy(agent.pp["obs_stats"][0]), params=npy(agent.model.params.flat), step=npy(agent.step_counter))
        res.append(f)
        assert agent.current_step == 2 * 64 * 48 and np.isfinite(info["actor_loss"])
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_whole_rollout_time_out_after_the_first_rollout_raises_at_the_update_readback():
    """Later rollouts are not read synchronously; their status words ride in the learner's read-back block and a raised
    flag makes PPO_Agent.update() fail loudly instead of training on a partly stale buffer."""
    from xuance_amd import ops
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(0)
    agent = PPO_Agent(make_config(64, 32, use_hip_graph=True), DeviceCartPoleVecEnv(64, seed=3))
    agent.rollout()
    assert agent.persist_status is not None and agent._persist_status_ok(agent.persist_status.tolist())
    agent.update()
    agent.rollout()
    agent.persist_status[0:1].fill_(1)                                # what a barrier time-out leaves behind
    with pytest.raises(ops.XrlError, match="xrl_rollout_cartpole_run"):
        agent.update()


def test_runner_surface_of_the_ppo_agent(tmp_path):
    """What xuance/engine/run_drl.py:101-203 calls on an agent: train -> save_model("final_train_model.pth") into
    model_dir_save (a seed_* run folder), a fresh agent's load_model(model_dir_load) finds it (drl_learner.py:95-157),
    test(test_episodes, test_envs=host vec env, close_envs) returns one score per finished episode, meta_data /
    distributed_training / current_step exist; get_actions has the reference's signature and output fields."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv, DummyVecEnv, NumpyCartPoleEnv
    n, T = 32, 32
    cfg = make_config(n, T, use_hip_graph=True, n_epochs=4, n_minibatch=4, model_dir=str(tmp_path / "models"), agent="PPO",
                      env_name="Classic Control", env_id="CartPole-v1")
    torch.manual_seed(0)
    a = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=3))
    assert a.distributed_training is False and a.model_dir_load == cfg.model_dir and a.meta_data["algo"] == "PPO"
    a.train(8 * T)
    a.save_model(model_name="final_train_model.pth")
    run_dirs = [d for d in (tmp_path / "models").iterdir()]
    assert len(run_dirs) == 1 and run_dirs[0].name.startswith("seed_1_") and (run_dirs[0] / "obs_rms.npy").exists()
    b = PPO_Agent(make_config(n, T, use_hip_graph=True, n_epochs=4, n_minibatch=4, model_dir=str(tmp_path / "models")),
                  DeviceCartPoleVecEnv(n, seed=3))
    loaded = b.load_model(b.model_dir_load)
    assert loaded == str(run_dirs[0])
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[k]), k
    for x, y in zip(a._obs_stats_tensors(), b._obs_stats_tensors()):
        assert torch.equal(x.float(), y.float())
    test_envs = DummyVecEnv([NumpyCartPoleEnv] * 4, env_seed=11)
    scores = b.test(test_episodes=6, test_envs=test_envs, close_envs=True)
    assert len(scores) >= 6 and all(8 <= s <= 500 for s in scores) and test_envs.closed
    assert b.logged[-1][1]["Test-Episode-Rewards/Mean-Score"] == float(np.mean(scores))
    # get_actions (on_policy.py:128-169): processed observations in, NumPy actions / values (/ log-probs) out
    obs = np.random.default_rng(0).standard_normal((5, 4)).astype(np.float32)
    out = b.get_actions(obs, deterministic=True, return_dists=True, return_logpi=True)
    sd = {k: npy(v) for k, v in b.model.state_dict().items()}
    from oracle import xrl_oracle as o
    logits, value = o.actor_critic_forward(sd, obs)
    assert np.array_equal(out.env_actions, logits.argmax(-1)) and out.env_actions.dtype == np.int64
    assert_close(out.values, value, 1e-5, "values")
    assert_close(out.distributions["logits"], logits, 1e-5, "logits")
    assert_close(out.log_probs, o.log_softmax(logits)[np.arange(5), out.env_actions], 1e-5, "log-probs")
    out = b.get_actions(obs, return_logpi=True)                       # stochastic draw
    assert set(out.env_actions.tolist()) <= {0, 1}
    assert_close(out.log_probs, o.log_softmax(logits)[np.arange(5), out.env_actions], 1e-5, "log-probs of the draw")
    b.finish()


@pytest.mark.parametrize("graph", [False, True])
def test_ppokl_agent_stores_the_old_distribution_and_adapts_its_coefficient(graph):
    """PPOKL_Agent (ppokl_agent.py:7-90) on the device CartPole: the vector step stores the old distribution's parameters
    next to old_logp (they must reproduce it), the update phase feeds them to PPOKL_Learner (loss mode 3), the coefficient
    moves within [0.1, 20] on the device across the chained minibatches, eager and as one graph alike."""
    from xuance_amd.agents import REGISTRY_Agents
    from xuance_amd.envs import DeviceCartPoleVecEnv
    n, T = 16, 32
    torch.manual_seed(0)
    cfg = make_config(n, T, n_epochs=2, n_minibatch=2, target_kl=0.01, kl_coef=1.0, use_hip_graph=graph)
    agent = REGISTRY_Agents["PPOKL"](cfg, DeviceCartPoleVecEnv(n, seed=2))
    p0 = agent.model.params.flat.clone()
    coefs = []
    for _ in range(3):
        agent.rollout()
        f = agent.memory.soa.fields
        logits, act, logp = npy(f["aux_old_a"]), npy(f["actions"]).astype(np.int64), npy(f["aux_old_logp"])
        lsm = logits - np.log(np.exp(logits - logits.max(-1, keepdims=True)).sum(-1, keepdims=True)) - logits.max(-1, keepdims=True)
        assert_close(np.take_along_axis(lsm, act[..., None], -1)[..., 0], logp, 1e-5, "old_logp from the stored logits")
        info = agent.update()
        assert set(info) == {"actor-loss", "critic-loss", "entropy", "learning_rate", "kl", "predict_value"}
        assert all(np.isfinite(v) for v in info.values()) and info["kl"] >= 0.0
        coefs.append(agent.learner.kl_coef)
        assert 0.1 <= coefs[-1] <= 20.0
    assert len(set(coefs)) > 1 or coefs[0] != 1.0                    # the schedule moved
    assert not torch.equal(agent.model.params.flat, p0)
    assert float(agent.learner.kl_coef_dev[0].item()) == agent.learner.kl_coef        # [current, used by the last loss]


def test_ppo_agent_on_atari_shape():
    """PPO on uint8 frame stacks (configs/ppo/atari.yaml: AC_CNN_Atari representation, HipOnPolicyBuffer_Atari = the
    reference's DummyOnPolicyBuffer_Atari, memory_tools.py:290-328) end to end on the device: the rollout stores the provider's
    frames as they are, values / log-probs in the buffer are the network's own outputs on those frames, truncated paths bootstrap
    from the next frames' values, and the update phase straight from the buffer equals PPO_Learner.update(**samples) on the same
    rows (gather of uint8 rows + the same launches)."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    from xuance_amd.nets import ActorCriticCNN
    from xuance_amd.memory import HipOnPolicyBuffer_Atari
    torch.manual_seed(0)
    n, T = 8, 16
    cfg = make_config(n, T, representation="AC_CNN_Atari", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                      fc_hidden_sizes=[512], actor_hidden_size=[], critic_hidden_size=[], activation="relu", n_epochs=1,
                      n_minibatch=2, use_obsnorm=False, use_rewnorm=True, learning_rate=2.5e-4, gamma=0.99, use_hip_graph=False)
    env = SyntheticAtariVecEnv(n, seed=5, max_episode_steps=9)
    agent = PPO_Agent(cfg, env)
    assert isinstance(agent.model, ActorCriticCNN) and isinstance(agent.memory, HipOnPolicyBuffer_Atari) and agent.frames
    assert agent.memory.soa.fields["observations"].dtype == torch.uint8
    p0 = agent.model.params.flat.clone()
    agent.rollout()
    torch.cuda.synchronize()
    f = agent.memory.soa.fields
    obs = f["observations"].view(T, n, -1)
    seg = npy(f["seg"])
    assert int(obs.max()) > 0 and (seg[:9] & 1).any(0).all() and (seg[T - 1] & 1).all()   # every env: a path end within 9 steps (cut-off
    #                                                                                      or the provider's random terminations); buffer end
    # the stored values / log-probs are the network's outputs on the stored frames (initial parameters)
    heads = agent.model.forward(obs.reshape(T * n, -1), T * n, keep=False)[:T * n].clone()
    A = agent.model.action_dim
    assert_close(npy(f["values"]).reshape(-1), npy(heads[:, A]), 1e-5, "values")
    lp = torch.log_softmax(heads[:, :A], -1).gather(1, f["actions"].view(-1, 1).long())[:, 0]
    assert_close(npy(f["aux_old_logp"]).reshape(-1), npy(lp), 1e-5, "old_logp", scale=float(lp.abs().max()))
    acts = npy(f["actions"])
    assert set(np.unique(acts)) <= set(range(A))
    # update phase from the buffer == update(**samples) on the same minibatches
    idx = np.stack([np.random.default_rng(3).permutation(n * T)]).reshape(2, -1)
    agent.set_indices(idx)
    info = agent.update()
    pa = agent.model.params.flat.clone()
    agent.model.params.flat.copy_(p0)
    twin = PPO_Agent(cfg, SyntheticAtariVecEnv(n, seed=5, max_episode_steps=9))
    twin.model.params.flat.copy_(p0)
    fields = {k: npy(v) for k, v in f.items()}
    for k in range(2):
        env_i, t_i = np.divmod(idx[k], T)
        adv = fields["advantages"][t_i, env_i]
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        info2 = twin.learner.update(obs=fields["observations"][t_i, env_i], actions=fields["actions"][t_i, env_i],
                                    returns=fields["returns"][t_i, env_i], values=fields["values"][t_i, env_i], advantages=adv,
                                    aux_batch={"old_logp": fields["aux_old_logp"][t_i, env_i]}, batch_size=len(env_i))
    for key in ("critic_loss", "entropy", "predict_value"):
        assert_close(info[key], info2[key], 1e-5, key)
    # (the advantage statistics are float64 on the device, float32 NumPy here; Adam turns differences at the level of its eps into
    #  steps of either sign for the few entries whose gradient sits there: compared in the 2-norm over all 3.36 M parameters)
    d = (pa - twin.model.params.flat).double().norm().item()
    moved = (pa - p0).double().norm().item()
    assert moved > 1e-3 and d <= 1e-2 * moved, (d, moved)


def test_ppo_atari_rollout_as_one_graph_equals_the_eager_rollout():
    """The frame provider alternates between two observation buffers (period 2) and numbers its steps from a device counter inside a
    captured rollout (step_device(offset=t) + advance(T)): a captured rollout of an even horizon replays on the same addresses and
    draws the same random streams as the eager loop -- two consecutive rollouts, every buffer field and the provider's state bit-equal."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n, T = 4, 6
    out = []
    for graph in (False, True):
        torch.manual_seed(0)
        cfg = make_config(n, T, representation="AC_CNN_Atari", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                          fc_hidden_sizes=[512], actor_hidden_size=[], critic_hidden_size=[], activation="relu", n_epochs=1,
                          n_minibatch=2, use_obsnorm=False, use_rewnorm=True, learning_rate=2.5e-4, gamma=0.99, use_hip_graph=graph)
        agent = PPO_Agent(cfg, SyntheticAtariVecEnv(n, seed=5, max_episode_steps=5))
        snaps = []
        for _ in range(2):
            agent.rollout()
            torch.cuda.synchronize()
            f = agent.memory.soa.fields
            snaps.append({k: npy(v) for k, v in f.items()} | {"env_obs": npy(agent.envs.buf_obs), "env_steps": npy(agent.envs.steps)})
        assert (agent._rollout_graph is not None) == graph
        out.append(snaps)
    for a, b in zip(*out):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("graph", [False, True])
def test_ppo_atari_acting_tail_in_one_launch_equals_the_launches(graph):
    """xrl_ppo_act_tail (round 6): behind the hidden layer's split-K product of a vector step on frame stacks -- epilogue, logits + value,
    sample / log-prob / value / bootstrap value, the previous step's bookkeeping, the copy of the frames into their buffer slot -- as ONE
    launch against the five launches it replaces (config.use_frame_act_tail: False): two consecutive rollouts with episode ends of both
    kinds inside, every buffer field (actions, log-probs, values, bootstrap values, processed rewards, flags, the stored frames,
    advantages, returns), the return statistics and the provider's state bit-equal; also with the reward normalisation on."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n, T = 8, 12
    out = []
    for tail in (True, False):
        torch.manual_seed(0)
        cfg = make_config(n, T, representation="AC_CNN_Atari", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                          fc_hidden_sizes=[512], actor_hidden_size=[], critic_hidden_size=[], activation="relu", n_epochs=1,
                          n_minibatch=2, use_obsnorm=False, use_rewnorm=True, learning_rate=2.5e-4, gamma=0.99, use_hip_graph=graph,
                          use_frame_act_tail=tail)
        agent = PPO_Agent(cfg, SyntheticAtariVecEnv(n, seed=5, max_episode_steps=5, p_term=0.05))
        snaps = []
        for _ in range(2):
            agent.rollout()
            torch.cuda.synchronize()
            assert bool(getattr(agent, "_ftail_ok", False)) == tail
            f = agent.memory.soa.fields
            snaps.append({k: npy(v) for k, v in f.items()} | {"env_obs": npy(agent.envs.buf_obs), "env_steps": npy(agent.envs.steps),
                                                               "ret_mean": npy(agent.ret_mean), "ret_var": npy(agent.ret_var),
                                                               "ret_count": npy(agent.ret_count), "returns_track": npy(agent.returns)})
        out.append(snaps)
    for a, b in zip(*out):
        assert (a["seg"] & 1).any() and a["terminals"].any() and int(a["observations"].max()) > 0
        for k in a:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name,n,T", [("DeviceAcrobotVecEnv", 70, 12), ("DeviceMountainCarVecEnv", 16, 10), ("DevicePendulumVecEnv", 130, 8),
                                      ("DeviceCartPoleVecEnv", 64, 10)])
def test_device_act_tail_in_one_launch_equals_the_launches(name, n, T, graph):
    """xrl_act_tail + xrl_post_norm (round 6): the general path's vector step on a device env -- normalise + store, three grouped products,
    xrl_policy_sample, the env's step, xrl_rollout_poststep: seven launches -- as four: [the previous step's bookkeeping + statistics /
    normalisation / store], the two hidden stages, [heads + sampling + env step] (config.use_device_act_tail; off by default: no faster) --
    and as six (config.use_post_norm, the default: only the first merge) -- against the seven (both switches off).
    Three training iterations (rollout + update, so the parameters move between the rollouts) with episode ends of both kinds inside:
    every buffer field (observations, actions, log-probs, values, bootstrap values, processed rewards, flags, advantages, returns), the
    observation / return statistics, the env's state and counters and the parameters bit-equal; categorical and Gaussian heads, several
    workgroups with a ragged last one (70 / 130 envs at 16 per workgroup)."""
    import xuance_amd.envs as envs
    from xuance_amd.agents import PPO_Agent
    out = []
    for tail, post_norm in ((True, True), (False, True), (False, False)):     # (the last: the seven launches)
        torch.manual_seed(0)
        kw = dict(activation_action="tanh") if name == "DevicePendulumVecEnv" else {}
        env = getattr(envs, name)(n, seed=3)
        env.max_episode_steps = 7
        # (use_trunk_forward: False -- the three forms differ in how the launches AROUND the hidden stages are grouped; all of them on the
        #  layered float32 forward, which xrl_act_tail's form consumes stage by stage.  The one-launch acting pass has its own tests:
        #  tests/test_gpu_ppo.py::test_acting_pass_*, test_general_path_rollout_uses_the_one_launch_acting_pass)
        agent = PPO_Agent(make_config(n, T, use_hip_graph=graph, use_fused_rollout=False, use_device_act_tail=tail, use_post_norm=post_norm,
                                      use_trunk_forward=False, **kw), env)
        snaps = []
        for it in range(3):
            agent.train(T)
            torch.cuda.synchronize()
            assert (agent._device_tail() is not None) == tail and agent._post_norm_ok() == post_norm
            f = agent.memory.soa.fields
            snaps.append({k: npy(v) for k, v in f.items()} | {
                "env_obs": npy(env.buf_obs), "env_state": npy(env.state), "env_steps": npy(env.steps), "env_episodes": npy(env.episodes),
                "env_stats": npy(env.stats), "obs_mean": npy(agent.obs_mean), "obs_var": npy(agent.obs_var), "obs_count": npy(agent.obs_count),
                "ret_mean": npy(agent.ret_mean), "ret_var": npy(agent.ret_var), "ret_count": npy(agent.ret_count),
                "returns_track": npy(agent.returns), "params": npy(agent.model.params.flat),
                "heads_act_rows": npy(agent.model.plan.acts[len(agent.model.plan.widths) - 1][:n])})
        out.append(snaps)
    for other in out[:2]:
        for a, b in zip(other, out[2]):
            assert (a["seg"] & 1).any() and float(np.abs(a["observations"]).max()) > 0
            for k in a:
                assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name,dist", [("DevicePendulumVecEnv", "gaussian"), ("DeviceMountainCarVecEnv", "categorical"),
                                       ("DeviceAcrobotVecEnv", "categorical")])
def test_ppo_agent_on_the_other_classic_control_envs(name, dist):
    """configs/ppo/classic_control/{Pendulum-v1, MountainCar-v0, Acrobot-v1}.yaml end to end on the device: the env is a device env
    (xrl_classic_step), the rollout a captured graph of the layered step, the minibatch update the one-launch shared-trunk kernel
    (fused_eligible) -- two iterations run, parameters move, everything stays finite, and the graph path equals the eager path
    bit for bit (same launches, same Philox streams)."""
    import xuance_amd.envs as envs
    from xuance_amd.agents import PPO_Agent
    n, T = 16, 32
    out = []
    for graph in (False, True):
        torch.manual_seed(0)
        kw = dict(activation_action="tanh") if dist == "gaussian" else {}
        agent = PPO_Agent(make_config(n, T, use_hip_graph=graph, **kw), getattr(envs, name)(n, seed=3, max_episode_steps=20))
        assert agent.model.dist == dist and agent.learner.fused_eligible(agent.memory)
        p0 = agent.model.params.flat.clone()
        infos = [agent.train(T)]
        # acting on MORE rows than the loops use grows the dense workspaces (Plan.ensure reallocates): the captured rollout and
        # update graphs hold the old pointers and must be captured again, not replayed (they were replayed into freed memory --
        # the optimiser's barrier words, as it happened -- until round 3 sized the workspaces up front and added this check)
        agent.get_actions(np.zeros((600, agent.obs_dim), np.float32))
        infos += [agent.train(T) for _ in range(2)]
        torch.cuda.synchronize()
        assert all(np.isfinite(v) for i in infos for v in i.values() if isinstance(v, float))
        assert not torch.equal(agent.model.params.flat, p0)
        ep, score, length = agent.envs.episode_stats()
        assert ep >= n and 0 < length <= 20
        out.append(npy(agent.model.params.flat))
    assert np.array_equal(out[0], out[1])


def test_loop_callbacks_of_the_device_loops():
    """xuance/common/callback.py:31-58 in the device loops: on_train_epochs_end after every update phase and on_train_step_end once
    per rollout (kwargs steps = horizon) by default; with config.per_step_callbacks the rollout runs as per-step launches and
    on_train_step / on_train_step_end fire per vector step with the buffer's slot as device tensors (ppo_agent.py:123-126,179-180) --
    same rollout data either way.  The DQN loop is a host loop already: its hooks fire per vector step (off_policy.py:221-269)."""
    from xuance_amd.agents import PPO_Agent, DQN_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv

    class Rec:
        def __init__(self):
            self.calls = []

        def on_update_start(self, it, **kw):
            return {}

        def on_update_end(self, it, **kw):
            return {}

        def on_train_step(self, step, **kw):
            self.calls.append(("step", step, tuple(kw["obs"].shape), kw["obs"].is_cuda))

        def on_train_step_end(self, step, **kw):
            self.calls.append(("step_end", step, kw.get("steps")))

        def on_train_epochs_end(self, step, **kw):
            self.calls.append(("epochs_end", step, sorted(kw["update_info"])[:1]))
    n, T = 32, 8
    fields = []
    for per_step in (False, True):
        torch.manual_seed(0)
        cb = Rec()
        agent = PPO_Agent(make_config(n, T, per_step_callbacks=per_step, use_hip_graph=True), DeviceCartPoleVecEnv(n, seed=2), cb)
        agent.train(2 * T)
        torch.cuda.synchronize()
        fields.append({k: npy(v) for k, v in agent.memory.soa.fields.items()})
        kinds = [c[0] for c in cb.calls]
        assert kinds.count("epochs_end") == 2
        if per_step:
            assert kinds.count("step") == 2 * T and kinds.count("step_end") == 2 * T
            steps = [c[1] for c in cb.calls if c[0] == "step"]
            assert steps == [t * n for t in range(2 * T)] and cb.calls[0][2] == (n, 4) and cb.calls[0][3]
            assert [c[1] for c in cb.calls if c[0] == "step_end"] == [(t + 1) * n for t in range(2 * T)]
        else:
            assert kinds.count("step") == 0 and [c for c in cb.calls if c[0] == "step_end"] == [("step_end", n * T, T), ("step_end", 2 * n * T, T)]
    for k in fields[0]:
        assert np.array_equal(fields[0][k], fields[1][k]), k        # per-step launches == the whole-rollout launch, bit for bit
    cb = Rec()
    cfg = Namespace(representation_hidden_size=[64], q_hidden_size=[64], activation="relu", seed=1, parallels=16, running_steps=10 ** 5,
                    buffer_size=16 * 64, batch_size=32, learning_rate=1e-3, gamma=0.99, start_greedy=0.5, end_greedy=0.05,
                    decay_step_greedy=10000, sync_frequency=50, training_frequency=16, start_training=64, use_grad_clip=False,
                    grad_clip_norm=0.5, use_obsnorm=False, use_rewnorm=False, distributed_training=False, device="cuda", model_dir="/tmp/x")
    dqn = DQN_Agent(cfg, DeviceCartPoleVecEnv(16, seed=3), cb)
    dqn.train(12)
    kinds = [c[0] for c in cb.calls]
    assert kinds.count("step") == 12 and kinds.count("step_end") == 12 and kinds.count("epochs_end") >= 5
