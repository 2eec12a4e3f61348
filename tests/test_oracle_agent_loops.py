"""CPU: the ORACLE's restatement of the agent loops (oracle/xrl_oracle.py) pinned to runs of the REFERENCE's own agents
(tests/golden/agent_*.npz, oracle/make_golden_agents.py: the unmodified PPO_Agent.train / DQN_Agent.train / QMIX_Agents.train
recorded through their callback hooks).  Inputs taken from the fixture: the simulators' outputs, the reference's random decisions
(sampled actions, exploration coins, random actions, sample indices).  Everything the loop COMPUTES -- running statistics,
normalised observations, values / log-probs, processed rewards, path closing, GAE, epsilon, greedy actions, update triggers, ring
positions, the parameters after every update phase -- is recomputed by the oracle and compared with the reference's.  The device
loops are compared with the same fixtures in tests/test_gpu_agent_replay.py."""
import numpy as np
import pytest

from conftest import load_golden, sub, assert_close, ChainCheck


def a2c_names(keys):
    """The reference's ActorCritic keys (one representation per head: actor.representation.model.*, actor.actor_head.logits.*, critic.*)
    -> the oracle's layer-chain names (actor.logits.<2i>, critic.values.<2i>: the representation's layers first)."""
    out = {}
    for head, hk, ok in (("actor", "actor_head.logits", "actor.logits"), ("critic", "critic_head.values", "critic.values")):
        nrep = len({k.split(".")[3] for k in keys if k.startswith(f"{head}.representation.model.")})
        for k in keys:
            if k.startswith(f"{head}.representation.model."):
                i, suf = k.split(".")[3], k.split(".")[4]
                out[k] = f"{ok}.{int(i)}.{suf}"
            elif k.startswith(f"{head}.{hk}."):
                i, suf = k.split(".")[3], k.split(".")[4]
                out[k] = f"{ok}.{int(i) + 2 * nrep}.{suf}"
    return out


@pytest.mark.parametrize("kind", ["categorical", "categorical40", "gaussian", "gaussian40", "a2c"])
def test_ppo_agent_loop(oracle, kind):
    """ppo_agent.py:111-181 + core/on_policy.py:182-205: per vector step obs_rms.update(raw obs) -> normalise -> act -> env ->
    store (normalised obs, action, processed reward, value, TERMINATED flag, old_logp); buffer full: V(next_obs) under the CURRENT
    statistics closes every path (0 for terminated envs), the update phase runs, the buffer is cleared; after that the return
    tracker / ret_rms / per-env path closing of finished episodes (no-ops on the emptied buffer when both coincide).
    categorical: agent_ppo.npz (CartPole yaml); gaussian: agent_ppo_gaussian.npz (mujoco yaml: 17-256-256-{6, 1}, tanh on the mean,
    state-independent log_std -- actions are NOT rescaled or clipped on the way to the env, wrapper.py:19,90-91)."""
    o = oracle
    gauss, a2c = kind in ("gaussian", "gaussian40"), kind == "a2c"
    # a2c: agent_a2c.npz -- A2C_Agent on the generic loop (core/on_policy.py:232-300) with configs/a2c/classic_control/CartPole-v1.yaml:
    # the same path-closing rules with V(next_obs) from the critic's own representation, no old_logp in the buffer, A2C_Learner's loss
    g = load_golden({"a2c": "agent_a2c", "gaussian": "agent_ppo_gaussian", "gaussian40": "agent_ppo_gaussian_40", "categorical40": "agent_ppo_40",
                     "categorical": "agent_ppo"}[kind])
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, T, E, MB = (int(c[k]) for k in ("n_envs", "horizon_size", "n_epochs", "n_minibatch"))
    S, D = g["step/acts"].shape[0], g["raw_obs0"].shape[1]
    A = g["step/acts"].shape[2] if gauss else 2
    fw = dict(dist="gaussian", act="leaky_relu", activation_action="tanh") if gauss else {}
    names = a2c_names(list(sub(g, "init"))) if a2c else {k: k for k in sub(g, "init")}
    back = {v: k for k, v in names.items()}
    sd = {names[k]: v.copy() for k, v in sub(g, "init").items()}
    opt = o.AdamOracle(sd, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    ucfg = dict(vf_coef=c["vf_coef"], ent_coef=c["ent_coef"], clip_range=c["clip_range"], use_grad_clip=True, grad_clip_norm=c["grad_clip_norm"])
    if a2c:
        fw = dict(loss_kind="a2c")
    obs_rms, ret_rms = o.RunningMeanStdOracle((D,)), o.RunningMeanStdOracle(())
    returns = np.zeros(n, np.float32)
    buf = o.OnPolicyBufferOracle((D,), (A,) if gauss else (), n, T, gamma=c["gamma"], gae_lam=c["gae_lambda"])
    raw = g["raw_obs0"].copy()
    phase = 0
    for s in range(S):
        obs_rms.update(raw)
        obs_n = o.process_observation(raw, obs_rms, c["obsnorm_range"]).astype(np.float32)
        assert_close(obs_n, g["step/obs"][s], 1e-6, f"step {s}: normalised obs")
        afw = {k: v for k, v in fw.items() if k != "loss_kind"}
        head, value = o.actor_critic_forward(sd, obs_n, **afw)
        acts = g["step/acts"][s]                                                    # fixed input: what the reference sampled
        if gauss:
            assert_close(head, g["step/mu"][s], 1e-5, f"step {s}: mean of the action distribution", scale=1.0)
            assert_close(np.exp(sd["actor.log_std"]), g["step/std"][s].reshape(-1, A)[0], 1e-6, "std")
            # (the normals the GPU replay supplies reproduce these actions: x = mu + std z)
            z = ((acts - g["step/mu"][s]) / g["step/std"][s]).astype(np.float32)
            x, logp = o.gaussian_sample_reparam(head, sd["actor.log_std"], z)
            assert_close(x, acts, 1e-5, f"step {s}: actions from the supplied normals", scale=max(1.0, float(np.abs(acts).max())))
            logp = o.gaussian_sample_reparam(head, sd["actor.log_std"], ((acts - head) / np.exp(sd["actor.log_std"])).astype(np.float32))[1]
            assert_close(logp, g["step/logp"][s], 1e-5, f"step {s}: log-probs", scale=max(1.0, float(np.abs(g["step/logp"][s]).max())))
        else:
            probs = np.exp(o.log_softmax(head))
            assert_close(probs, g["step/probs"][s], 1e-5, f"step {s}: action probabilities")
            # (the uniforms the GPU replay supplies reproduce these actions through the inverse CDF)
            cdf = np.cumsum(g["step/probs"][s].astype(np.float64), -1)
            u = (cdf[np.arange(n), acts] - 0.5 * g["step/probs"][s][np.arange(n), acts]).astype(np.float32)
            assert np.array_equal(o.categorical_sample_icdf(head, u), acts)
            logp = o.log_softmax(head)[np.arange(n), acts]
            if not a2c:                                                              # (A2C's loop asks for no log-probs, on_policy.py:236)
                assert_close(logp, g["step/logp"][s], 1e-5, f"step {s}: log-probs")
        assert_close(value, g["step/vals"][s], 1e-5, f"step {s}: values", scale=max(1e-2, float(np.abs(g["step/vals"][s]).max())))
        next_obs, rew, term, trunc = g["step/next_obs"][s], g["step/rewards"][s], g["step/terminals"][s], g["step/truncations"][s]
        buf.store(obs_n, acts, o.process_reward(rew, ret_rms, c["rewnorm_range"]), value, term, {"old_logp": g["step/logp"][s] if gauss else logp})
        if buf.full:
            vals = o.actor_critic_forward(sd, o.process_observation(next_obs, obs_rms, c["obsnorm_range"]).astype(np.float32), **afw)[1]
            for i in range(n):
                buf.finish_path(0.0 if term[i] else vals[i], i)
            ref = sub(g, f"phase{phase}/buffer")
            assert np.array_equal(buf.actions, ref["actions"]) and np.array_equal(buf.terminals > 0, ref["terminals"] > 0)
            for k in ("observations", "rewards", "values", "returns"):
                assert_close(getattr(buf, k), ref[k], 1e-5, f"phase {phase}: buffer {k}")
            if not a2c:
                assert_close(buf.old_logp, ref["old_logp"], 1e-5, f"phase {phase}: buffer old_logp")
            assert_close(buf.advantages, ref["advantages"], 1e-5, f"phase {phase}: buffer advantages", scale=float(np.abs(ref["returns"]).max()))
            idx = g[f"phase{phase}/indices"]
            assert idx.shape == (E * MB, n * T // MB)
            for e in range(E):                                                      # each epoch's minibatches partition the buffer
                assert np.array_equal(np.sort(idx[e * MB:(e + 1) * MB].ravel()), np.arange(n * T))
            for k in range(E * MB):
                b = buf.sample(idx[k])
                info, grads = o.ppo_update(sd, opt, dict(obs=b["obs"], actions=b["actions"], returns=b["returns"], advantages=b["advantages"],
                                                         old_logp=b["aux_batch"]["old_logp"]), ucfg, **fw)
                ref_g = sub(g, f"phase{phase}/grad{k}")
                for name, rg in ref_g.items():
                    assert_close(info["clipped_grads"][names[name]], rg, 1e-5, f"phase {phase} update {k}: clipped gradient {name}")
            ri = sub(g, f"phase{phase}/info")
            assert_close(info["a_loss"], ri["actor-loss" if a2c else "actor_loss"], 1e-5, "actor_loss", scale=1.0)
            assert_close(info["c_loss"], ri["critic-loss" if a2c else "critic_loss"], 1e-5, "critic_loss")
            assert_close(info["e_loss"], ri["entropy"], 1e-5, "entropy")
            for name, rp in sub(g, f"phase{phase}/param").items():
                moved = float(np.abs(rp - g[f"init/{name}"]).max())
                assert_close(sd[names[name]], rp, 2e-4, f"phase {phase}: parameter {name} (relative to the distance it moved)", scale=moved)
            buf.clear()
            phase += 1
        returns = (c["gamma"] * returns + rew).astype(np.float32)
        raw = next_obs.copy()
        for i in range(n):
            if term[i] or trunc[i]:
                ret_rms.update(returns[i:i + 1])
                returns[i] = 0.0
                if term[i]:
                    buf.finish_path(0.0, i)
                else:
                    vals = o.actor_critic_forward(sd, o.process_observation(next_obs, obs_rms, c["obsnorm_range"]).astype(np.float32), **afw)[1]
                    buf.finish_path(vals[i], i)
                raw[i] = g["step/reset_obs"][s][i]
        assert_close(returns, g["step/returns_track"][s], 1e-5, f"step {s}: return tracker", scale=max(1.0, float(np.abs(g["step/returns_track"][s]).max())))
        assert_close(obs_rms.mean, g["step/obs_rms/mean"][s], 1e-5, "obs_rms.mean", scale=float(np.sqrt(g["step/obs_rms/var"][s]).max()))
        assert_close(obs_rms.var, g["step/obs_rms/var"][s], 1e-5, "obs_rms.var")
        assert_close(ret_rms.var, g["step/ret_rms/var"][s], 1e-5, "ret_rms.var")
        assert_close(obs_rms.count, g["step/obs_rms/count"][s], 1e-12, "obs_rms.count")
        assert_close(ret_rms.count, g["step/ret_rms/count"][s], 1e-12, "ret_rms.count")
    assert phase == S // T == (2 if kind == "categorical40" else 3)


def test_pg_agent_loop(oracle):
    """pg_agent.py:12-79 on the generic on-policy loop (core/on_policy.py:232-300), agent_pg.npz: actor-only policy (relu), the stored
    value of every step is 0 (on_policy.py:160), a path that is cut -- truncation or buffer end -- closes with the PROCESSED REWARD of
    its last step (get_terminated_values(next_obs, rewards) = _process_reward(rewards), pg_agent.py:66-79), a terminated one with 0;
    configs/pg/classic_control/CartPole-v1.yaml: use_gae False (returns = discounted reward sums, advantages = rewards + gamma v' - v
    with v = 0), no advantage normalisation, ONE update on the whole buffer per rollout (PG_Learner: -mean(returns log_prob) - ent_coef H)."""
    o = oracle
    g = load_golden("agent_pg")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, T = int(c["n_envs"]), int(c["horizon_size"])
    S = g["step/acts"].shape[0]
    sd = {k: v.copy() for k, v in sub(g, "init").items()}
    opt = o.AdamOracle(sd, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    obs_rms, ret_rms = o.RunningMeanStdOracle((4,)), o.RunningMeanStdOracle(())
    returns = np.zeros(n, np.float32)
    buf = o.OnPolicyBufferOracle((4,), (), n, T, gamma=c["gamma"], gae_lam=c["gae_lambda"], use_gae=False, use_advnorm=False)
    rep = lambda x: o.MLP(o.collect_seq(sd, "actor.actor_head.logits", "relu")).forward(
        o.MLP(o.collect_seq(sd, "actor.representation.model", "relu", last_act="relu")).forward(x))
    raw, phase = g["raw_obs0"].copy(), 0
    for s in range(S):
        obs_rms.update(raw)
        obs_n = o.process_observation(raw, obs_rms, c["obsnorm_range"]).astype(np.float32)
        assert_close(obs_n, g["step/obs"][s], 1e-6, f"step {s}: normalised obs")
        probs = np.exp(o.log_softmax(rep(obs_n)))
        assert_close(probs, g["step/probs"][s], 1e-5, f"step {s}: action probabilities")
        acts = g["step/acts"][s]
        assert not g["step/vals"][s].any()
        next_obs, rew, term, trunc = g["step/next_obs"][s], g["step/rewards"][s], g["step/terminals"][s], g["step/truncations"][s]
        rew_n = o.process_reward(rew, ret_rms, c["rewnorm_range"])
        buf.store(obs_n, acts, rew_n, np.zeros(n, np.float32), term, None)
        if buf.full:
            for i in range(n):
                buf.finish_path(0.0 if term[i] else rew_n[i], i)
            ref = sub(g, f"phase{phase}/buffer")
            assert np.array_equal(buf.actions, ref["actions"]) and np.array_equal(buf.terminals > 0, ref["terminals"] > 0)
            for k in ("observations", "rewards", "returns", "advantages"):
                assert_close(getattr(buf, k), ref[k], 1e-5, f"phase {phase}: buffer {k}", scale=float(np.abs(ref["returns"]).max()) if k == "advantages" else None)
            idx = g[f"phase{phase}/indices"]
            assert idx.shape == (1, n * T) and np.array_equal(np.sort(idx[0]), np.arange(n * T))
            b = buf.sample(idx[0])
            info, grads = o.pg_forward_backward(sd, dict(obs=b["obs"], actions=b["actions"], returns=b["returns"]), dict(ent_coef=c["ent_coef"]), act="relu")
            o.AdamOracle.clip_grad_norm_(grads, c["grad_clip_norm"])
            for name, rg in sub(g, f"phase{phase}/grad0").items():
                assert_close(grads[name], rg, 3e-5 if name.endswith("logits.2.bias") else 1e-5, f"phase {phase}: clipped gradient {name}")   # (the 2-element head bias: tests/test_oracle_vs_golden.py: test_pg_update)
            opt.step(grads)
            ri = sub(g, f"phase{phase}/info")
            assert_close(info["a_loss"], ri["actor-loss"], 1e-5, "actor-loss", scale=float(np.abs(info["log_prob"]).mean()))
            assert_close(info["e_loss"], ri["entropy"], 1e-5, "entropy")
            for name, rp in sub(g, f"phase{phase}/param").items():
                moved = float(np.abs(rp - g[f"init/{name}"]).max())
                assert_close(sd[name], rp, 2e-4, f"phase {phase}: parameter {name} (relative to the distance it moved)", scale=moved)
            buf.clear()
            phase += 1
        returns = (c["gamma"] * returns + rew).astype(np.float32)
        raw = next_obs.copy()
        for i in range(n):
            if term[i] or trunc[i]:
                ret_rms.update(returns[i:i + 1])
                returns[i] = 0.0
                buf.finish_path(0 if term[i] else o.process_reward(rew, ret_rms, c["rewnorm_range"])[i], i)   # (:279-283: AFTER ret_rms.update of this env)
                raw[i] = g["step/reset_obs"][s][i]
        assert_close(returns, g["step/returns_track"][s], 1e-5, f"step {s}: return tracker", scale=max(1.0, float(np.abs(g["step/returns_track"][s]).max())))
        assert_close(ret_rms.var, g["step/ret_rms/var"][s], 1e-5, "ret_rms.var")
    assert phase == 3


@pytest.mark.parametrize("kind", ["dummy", "atari", "subproc"])
def test_dqn_agent_loop(oracle, kind):
    """subproc (round 6): agent_dqn_subproc.npz -- the CartPole run behind the reference's SubprocVecEnv (40 vector steps): buf_obs is
    rebound there, the first stored observation is the one acted on (no alias, see `s == 0` below).
    atari: agent_dqn_atari.npz (configs/dqn/atari.yaml: uint8 frame stacks, Basic_CNN) -- the oracle has no convolutions, so the
    Q values are not recomputed there (the device replay does that); everything else of the loop is, plus the Atari rule: an env that
    terminated WITHOUT truncation keeps acting on its next observation (off_policy.py:240-242).
    core/off_policy.py:183-270 with dqn_agent.py:28-30: per vector step (obs_rms / normalisation are off in configs/dqn/*.yaml)
    greedy action of the eval network, the per-env coin `torch.rand(n) < e_greedy` against random actions (:138-141), env step, store
    (obs, action, reward, TERMINATED flag, next_obs) at the ring's write position; an update phase when `current_step >
    start_training and current_step % training_frequency == 0` (:228) on `np.random.choice` (env, step < size) pairs; then
    current_step += n_envs and the epsilon schedule (:119-127: recomputed while the PREVIOUS value is above end_greedy, so it
    undershoots the floor once -- the fixture ends at -0.0063)."""
    o = oracle
    atari, subproc = kind == "atari", kind == "subproc"
    g = load_golden("agent_dqn_atari" if atari else "agent_dqn_subproc" if subproc else "agent_dqn")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, S, B = int(c["n_envs"]), int(c["n_steps"]), int(c["batch_size"])
    sd = {k: v.copy() for k, v in sub(g, "init").items()}
    trainable = [k for k in sd if not k.startswith("target_")]
    opt = o.AdamOracle({k: sd[k] for k in trainable}, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    buf = o.OffPolicyBufferOracle(g["raw_obs0"].shape[1:], (), n, int(c["buffer_size"]), B, obs_dtype=g["raw_obs0"].dtype)
    eps_seq = o.egreedy_schedule(c["start_greedy"], c["end_greedy"], c["decay_step_greedy"], n, S + 1)
    raw = g["raw_obs0"].copy()
    cur, phase, updates, flips = 0, 0, 0, 0
    for s in range(S):
        assert eps_seq[s] == g["step/eps_acted"][s] and int(g["step/step_index"][s]) == cur
        if not atari:
            q = o.MLP(o.collect_seq(sd, "eval_Q_head.q_value", "relu")).forward(
                o.MLP(o.collect_seq(sd, "representation.model", "relu", last_act="relu")).forward(raw))
            greedy = q.argmax(-1)
            for e in np.flatnonzero(greedy != g["step/greedy"][s]):                 # an argmax may flip only on a tie at 1e-5
                assert abs(q[e, 0] - q[e, 1]) < 1e-5 * max(1.0, np.abs(q[e]).max()), (s, e, q[e])
                flips += 1
        acts = o.egreedy_select(g["step/greedy"][s], g["step/random_actions"][s], g["step/coin"][s], np.float32(eps_seq[s]))
        assert np.array_equal(acts, g["step/acts"][s]), f"step {s}: actions"
        next_obs, rew, term, trunc = g["step/next_obs"][s], g["step/rewards"][s], g["step/terminals"][s], g["step/truncations"][s]
        stored = raw
        if s == 0 and subproc:
            assert np.array_equal(g["step/obs"][0], raw) and not np.array_equal(raw, next_obs)       # (SubprocVecEnv: no alias)
        elif s == 0:
            # A quirk of the reference's FIRST vector step of a train() call, input data here: with normalisation off `obs` is still
            # the vector env's own buf_obs array (off_policy.py:184,187; agent.py:262-283 returns its argument), which
            # DummyVecEnv.step_wait overwrites in place (dummy_vec_env.py:74,88-93) -- what the loop stores as "obs" of that step
            # is already the NEXT observation.  (SubprocVecEnv rebinds buf_obs, subproc_vec_env.py:117: no alias there; from the
            # second step on `obs` is a deepcopy, off_policy.py:238.)  The policy acted on the true observation.
            assert np.array_equal(g["step/obs"][0], next_obs) and not np.array_equal(raw, next_obs)
            stored = next_obs
        else:
            assert np.array_equal(raw, g["step/obs"][s])
        buf.store(stored, acts, rew, term, next_obs)
        if cur > c["start_training"] and cur % int(c["training_frequency"]) == 0:
            assert int(g[f"phase{phase}/at_step"]) == s
            env_c, step_c = g[f"phase{phase}/indices"][0]
            assert step_c.max() < buf.size
            updates += 1
            if not atari:
                info, grads = o.dqn_forward_backward(sd, buf.sample_at(env_c, step_c), dict(gamma=c["gamma"]))
                for name, rg in sub(g, f"phase{phase}/grad0").items():
                    assert_close(grads[name], rg, 1e-5, f"phase {phase}: gradient {name}")
                assert_close(info["loss"], g[f"phase{phase}/info/Qloss"], 1e-5, "Qloss")
                opt.step(grads)
                if updates % int(c["sync_frequency"]) == 0:
                    o.dqn_copy_target(sd)
                for name, rp in sub(g, f"phase{phase}/param").items():
                    moved = float(np.abs(rp - g[f"init/{name}"]).max()) or 1.0
                    assert_close(sd[name], rp, 2e-4, f"phase {phase}: parameter {name} (relative to the distance it moved)", scale=moved)
            assert updates == int(g[f"phase{phase}/iterations"])
            phase += 1
        restart = trunc if atari else (term | trunc)                                  # (Atari mode: off_policy.py:240-242)
        raw = np.where(restart.reshape((n,) + (1,) * (next_obs.ndim - 1)), g["step/reset_obs"][s], next_obs)
        cur += n
        assert int(g["step/current_step"][s]) == cur and eps_seq[s + 1] == g["step/eps_after"][s]
        assert buf.ptr == int(g["step/ptr"][s]) and buf.size == int(g["step/size"][s])
    assert phase == int(g["n_phases"]) and flips <= 2
    fb = sub(g, "final_buffer")
    for k in ("observations", "next_observations", "actions", "rewards"):
        assert np.array_equal(getattr(buf, k), fb[k]), k
    assert np.array_equal(buf.terminals > 0, fb["terminals"] > 0)
    assert atari or any(k.startswith("target_") and not np.array_equal(sd[k], g[f"init/{k}"]) for k in sd)    # target syncs happened


def stored_state_rule(cur_state, done_prev):
    """off_policy_marl.py:395 (`state = info[i]["reset_state"]` replaces the whole list) + store_experience :151-153 + the buffer's
    per-env write (memory_tools_marl.py:731-740): after a vector step in which envs finished, what is stored as `state` of the NEXT
    step is the reset state of the last finished env, in every env's row."""
    if done_prev is None or not done_prev.any():
        return cur_state
    last = int(np.flatnonzero(done_prev)[-1])
    return np.broadcast_to(cur_state[last], cur_state.shape).copy()


@pytest.mark.parametrize("algo", ["qmix", "vdn", "iql"])
def test_qmix_ff_agent_loop(oracle, algo):
    """algo vdn / iql: VDN_Agents / IQL_Agents through the same loop (agent_{vdn,iql}_ff.npz): no global state is stored, the sum mixer /
    independent TD, and IQL's epsilon decays by (start - end) / decay_step_greedy per env step (iql_agents.py:37).
    core/off_policy_marl.py:358-424 with qmix_agents.py:40: per vector step the masked greedy actions of the shared Q network,
    ONE exploration coin for the whole step (:236: every agent of every env then takes a random AVAILABLE action), env step, store;
    an update phase of n_epochs updates when `current_step >= start_training and current_step % training_frequency == 0` (:376);
    epsilon = start - delta * current_step with delta = (start - end) / (decay_step_greedy / n_envs), clamped to end_greedy one
    step after it first falls below (:197-204).  Reference quirks that are INPUT DATA here: the first stored obs / avail_actions
    of a train() call are the vector env's lists after the step (aliases of DummyVecMultiAgentEnv.buf_obs / buf_avail_actions,
    off_policy_marl.py:358-359 with dummy_vec_maenv.py:74-75); the stored state after an episode end (stored_state_rule)."""
    o = oracle
    g = load_golden(f"agent_{algo}_ff")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, S, N, A, B, E = (int(c[k]) for k in ("n_envs", "n_steps", "n_agents", "n_actions", "batch_size", "n_epochs"))
    global_state = bool(g["uses_global_state"])
    assert global_state == (algo == "qmix")
    sd = {k: v.copy() for k, v in sub(g, "init").items()}
    trainable = [k for k in sd if not k.startswith("target_")]
    opt = o.AdamOracle({k: sd[k] for k in trainable}, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    buf = o.MarlBufferOracle(n, int(c["buffer_size"]) // n, N, 30, A, 48)
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))      # (parameters: within what gradients agreeing at 1e-5 allow)
    init = sub(g, "init")
    ocfg = dict(gamma=c["gamma"], double_q=True, use_actions_mask=True, mixer=algo)
    ipre = "shared/" if algo == "iql" else ""                                        # (IQL's info keys carry the group, iql_learner.py:128-131)
    delta = (c["start_greedy"] - c["end_greedy"]) / ((c["decay_step_greedy"] / n) if algo != "iql" else c["decay_step_greedy"])
    pe = "individual_q_networks.shared"
    obs, avail, state = g["acted_obs0"], g["acted_avail0"], g["acted_state0"]
    eps, cur, phase, updates, done_prev, ties = c["start_greedy"], 0, 0, 0, None, 0
    for s in range(S):
        assert eps == g["step/eps_acted"][s] and cur == int(g["step/step_index"][s])
        assert np.array_equal(avail, g["step/acted_avail"][s])
        h = o.MLP(o.collect_seq(sd, f"{pe}.representation.obs_representation.model", "relu", last_act="relu")).forward(obs.reshape(n * N, -1))
        q = o.MLP(o.collect_seq(sd, f"{pe}.critic_head.q_value", "relu")).forward(h)
        greedy = np.where(avail.reshape(n * N, A) > 0, q, -1e10).argmax(-1)                 # value_factorization.py:87-90
        ref_greedy = g["step/greedy"][s].reshape(-1)
        for r in np.flatnonzero(greedy != ref_greedy):
            assert abs(q[r, greedy[r]] - q[r, ref_greedy[r]]) < 1e-5 * max(1.0, np.abs(q[r]).max()), (s, r)
            ties += 1
        acts = g["step/acts"][s]
        if g["step/coin"][s] < eps:
            assert (avail.reshape(-1, A)[np.arange(n * N), acts.reshape(-1)] > 0).all()     # random AVAILABLE actions
        else:
            assert np.array_equal(acts, g["step/greedy"][s])
        st_obs, st_avail = (obs, avail) if s > 0 else (g["step/next_obs"][0], g["step/next_avail"][0])   # (the alias, see docstring)
        assert np.array_equal(g["step/stored_obs"][s], st_obs) and np.array_equal(g["step/stored_avail"][s], st_avail)
        st_state = stored_state_rule(state, done_prev) if global_state else np.zeros_like(state)
        assert np.array_equal(g["step/stored_state"][s], st_state), f"step {s}: stored state"
        buf.store(obs=st_obs, actions=acts, obs_next=g["step/next_obs"][s], rewards=g["step/rewards"][s], terminals=g["step/terminals"][s],
                  agent_mask=g["step/agent_mask"][s], state=st_state, state_next=g["step/next_state"][s], avail_actions=st_avail > 0,
                  avail_actions_next=g["step/next_avail"][s] > 0)
        if cur >= c["start_training"] and cur % int(c["training_frequency"]) == 0:
            assert int(g[f"phase{phase}/at_step"]) == s
            for e in range(E):
                env_c, step_c = g[f"phase{phase}/indices"][e]
                assert step_c.max() < buf.size
                b = buf.sample(env_c, step_c)
                info, grads = o.qmix_forward_backward(sd, b, ocfg, group="shared")
                for name, rg in sub(g, f"phase{phase}/grad{e}").items():
                    assert_close(grads[name], rg, 1e-5, f"phase {phase} update {e}: gradient {name}")
                ri = sub(g, f"phase{phase}/info{e}")
                assert_close(info["loss"], ri[ipre + "loss_Q"], 1e-5, "loss_Q")
                opt.step(grads)
                chain.step(sub(g, f"phase{phase}/grad{e}"))
                updates += 1
                if updates % int(c["sync_frequency"]) == 0:
                    o.qmix_copy_target(sd)
            assert updates == int(g[f"phase{phase}/iterations"])
            ref_p = sub(g, f"phase{phase}/param")
            if ref_p:
                chain.check({k: sd[k] for k in trainable}, {k: ref_p[k] for k in trainable}, init, what=f"phase {phase} param")
            phase += 1
        done_prev = g["step/done"][s]
        d3, d2 = done_prev[:, None, None], done_prev[:, None]
        obs = np.where(d3, g["step/reset_obs"][s], g["step/next_obs"][s])
        avail = np.where(d3, g["step/reset_avail"][s], g["step/next_avail"][s])
        state = np.where(d2, g["step/reset_state"][s], g["step/next_state"][s])
        cur += n
        eps = c["start_greedy"] - delta * cur if eps > c["end_greedy"] else c["end_greedy"]
        assert eps == g["step/eps_after"][s] and cur == int(g["step/current_step"][s])
        assert buf.ptr == int(g["step/ptr"][s]) and buf.size == int(g["step/size"][s])
    assert phase == int(g["n_phases"]) and ties <= 3
    fb = sub(g, "final_buffer")
    assert ("state" in fb) == global_state
    for k, v in fb.items():
        mine = buf.data[k]
        assert np.array_equal(np.asarray(mine, np.float32), np.asarray(v, np.float32).reshape(mine.shape)), f"final buffer field {k}"


def test_qmix_rnn_agent_loop(oracle):
    """core/off_policy_marl.py:334-356 + run_episodes :426-546 with recurrent agents (configs/qmix/sc2/3m.yaml, masks off): train()
    alternates run_episodes(n_envs) -- envs reset, staging rows cleared, GRU state zero; per vector step the greedy actions of the
    recurrent Q network on the carried state, ONE exploration coin, env step, store at each env's own episode step; a finished env
    closes its episode into the ring (terminal obs / state at slot episode_step), gets its GRU rows zeroed, and `current_step +=
    episode_step` followed by the epsilon update PER FINISHED ENV (:532-534) -- with n_epochs updates on `np.random.choice(size,
    batch)` episodes once current_step >= start_training.  The stored state follows stored_state_rule inside a call.  Which GRU
    rows a finished env zeroes is the reference's own rule (below): with any other rule the greedy actions of later steps differ."""
    o = oracle
    g = load_golden("agent_qmix_rnn")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, N, A, T, B, E = (int(c[k]) for k in ("n_envs", "n_agents", "n_actions", "max_episode_steps", "batch_size", "n_epochs"))
    sd = {k: v.copy() for k, v in sub(g, "init").items()}
    init = sub(g, "init")
    trainable = [k for k in sd if not k.startswith("target_")]
    opt = o.AdamOracle({k: sd[k] for k in trainable}, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    chain = ChainCheck(c["learning_rate"], total_iters=int(c["total_iters"]))
    buf = o.EpisodeBufferOracle(n, int(c["buffer_size"]), T, N, 30, A, 48)
    ocfg = dict(gamma=c["gamma"], double_q=True, use_actions_mask=False, agent_grad=False)   # (the unmodified reference: only the mixer trains, iql_learner.py:58)
    delta = (c["start_greedy"] - c["end_greedy"]) / (c["decay_step_greedy"] / n)
    eps, cur, s, phase, updates, ties = c["start_greedy"], 0, 0, 0, 0, 0
    n_calls = int(g["n_resets"])
    for call in range(n_calls):
        assert int(g[f"reset{call}/at"]) == s
        obs, state = g[f"reset{call}/obs"], g[f"reset{call}/state"]
        buf.clear_episodes()
        pe = "individual_q_networks.shared"
        rp = f"{pe}.representation.obs_representation"
        fc, qh = o.MLP(o.collect_seq(sd, f"{rp}.mlp", "relu", last_act="relu")), o.MLP(o.collect_seq(sd, f"{pe}.critic_head.q_value", "relu"))
        gw = [sd[f"{rp}.rnn.{k}_l0"] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        h = np.zeros((n * N, gw[1].shape[1]), np.float32)                            # init_rnn_states (value_factorization.py:151-159)
        done_prev, episodes = None, 0
        while episodes < n:
            assert int(g["step/call"][s]) == call and eps == g["step/eps_acted"][s] and cur == int(g["step/current_step_before"][s])
            assert np.array_equal(obs, g["step/acted_obs"][s])
            hs, _ = o.gru_forward(fc.forward(obs.reshape(n * N, -1))[:, None], h, *gw)   # one step on the carried state (rnn.py:52-77)
            h = hs[:, -1]
            q = qh.forward(h)
            greedy = q.argmax(-1).reshape(n, N)
            for e, a in zip(*np.nonzero(greedy != g["step/greedy"][s])):
                qq = q[e * N + a]
                assert abs(qq[greedy[e, a]] - qq[g["step/greedy"][s][e, a]]) < 1e-5 * max(1.0, np.abs(qq).max()), (s, e, a)
                ties += 1
            acts = g["step/acts"][s]
            if not g["step/coin"][s] < eps:
                assert np.array_equal(acts, g["step/greedy"][s])
            st_state = stored_state_rule(state, done_prev)
            assert np.array_equal(g["step/stored_state"][s], st_state), f"step {s}: stored state"
            es = g["step/episode_step"][s]
            buf.store(es - 1, obs=obs, actions=acts, rewards=g["step/rewards"][s], terminals=g["step/terminals"][s],
                      agent_mask=g["step/agent_mask"][s], state=st_state)
            done_prev = g["step/done"][s]
            for i in np.flatnonzero(done_prev):
                episodes += 1
                buf.finish_path(i, int(es[i]), g["step/next_obs"][s][i], g["step/next_state"][s][i], 0.0)
                # init_rnn_states_item(i_env=i) (:504-505) zeroes FLATTENED row i of the [n_envs * n_agents, H] state (batch_index =
                # [i_env], value_factorization.py:161-167, rnn.py:86-92) -- agent i % N of env i // N, not env i's agents
                h[i] = 0.0
                cur += int(es[i])
                eps = c["start_greedy"] - delta * cur if eps > c["end_greedy"] else c["end_greedy"]    # :197-204, per finished env
            d3, d2 = done_prev[:, None, None], done_prev[:, None]
            obs = np.where(d3, g["step/reset_obs"][s], g["step/next_obs"][s])
            state = np.where(d2, g["step/reset_state"][s], g["step/next_state"][s])
            s += 1
        assert s == int(g["call/n_steps"][call]) and cur == int(g["call/current_step"][call]) and eps == g["call/eps"][call]
        assert buf.ptr == int(g["call/ptr"][call]) and buf.size == int(g["call/size"][call])
        if cur >= c["start_training"]:
            assert int(g[f"phase{phase}/after_call"]) == call
            for e in range(E):
                ii = g[f"phase{phase}/indices"][e]
                assert ii.max() < buf.size
                d = buf.sample(ii)
                b = dict(obs=d["obs"].transpose(0, 2, 1, 3), actions=d["actions"].transpose(0, 2, 1), rewards=d["rewards"].transpose(0, 2, 1),
                         terminals=d["terminals"].transpose(0, 2, 1), agent_mask=d["agent_mask"].transpose(0, 2, 1),
                         avail_actions=np.ones((B, N, T + 1, A), np.float32), state=d["state"], filled=d["filled"])
                info, grads = o.qmix_rnn_forward_backward(sd, b, ocfg)
                for name, rg in sub(g, f"phase{phase}/grad{e}").items():
                    assert_close(grads[name], rg, 1e-5, f"phase {phase} update {e}: gradient {name}")
                assert_close(info["loss"], g[f"phase{phase}/info{e}/loss_Q"], 1e-5, "loss_Q")
                opt.step(grads)
                chain.step(sub(g, f"phase{phase}/grad{e}"))
                updates += 1
                if updates % int(c["sync_frequency"]) == 0:
                    o.qmix_copy_target(sd)
            assert updates == int(g[f"phase{phase}/iterations"])
            ref_p = sub(g, f"phase{phase}/param")
            chain.check({k: sd[k] for k in trainable}, {k: ref_p[k] for k in trainable}, init, what=f"phase {phase} param")
            phase += 1
    assert phase == int(g["n_phases"]) and s == g["step/acts"].shape[0] and ties <= 3
    for k, v in sub(g, "final_buffer").items():
        assert np.array_equal(np.asarray(buf.data[k], np.float32), np.asarray(v, np.float32).reshape(buf.data[k].shape)), f"ring field {k}"


def test_perdqn_agent_loop(oracle):
    """perdqn_agent.py:43-107 (agent_perdqn.npz): DQN's acting / store, then per update phase `memory.sample(PER_beta)` -- batch / n_envs
    stratified proportional draws per env from the sum tree (memory_tools.py:542-565; the recorded `random.random()` uniforms are the
    input), importance weights (recorded, not used by the reference's loss, perdqn_learner.py:49) -- `learner.update` (DQN's arithmetic),
    `update_priorities(step_choices, |TD error|)` (:586-597), `PER_beta += (1 - PER_beta0) / train_steps`; epsilon by the agent's OWN rule:
    minus delta per vector step while above end_greedy (:104-105)."""
    o = oracle
    g = load_golden("agent_perdqn")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, S, B = int(c["n_envs"]), int(c["n_steps"]), int(c["batch_size"])
    alpha, beta0 = g["per_cfg"].tolist()
    sd = {k: v.copy() for k, v in sub(g, "init").items()}
    trainable = [k for k in sd if not k.startswith("target_")]
    opt = o.AdamOracle({k: sd[k] for k in trainable}, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    buf = o.OffPolicyBufferOracle((4,), (), n, int(c["buffer_size"]), B)
    per = o.PerBufferOracle(n, int(c["buffer_size"]) // n, B, alpha)
    delta = (c["start_greedy"] - c["end_greedy"]) / (c["decay_step_greedy"] / n)
    raw = g["raw_obs0"].copy()
    eps, beta, cur, phase, updates = c["start_greedy"], beta0, 0, 0, 0
    for s in range(S):
        assert eps == g["step/eps_acted"][s] and cur == int(g["step/step_index"][s])
        acts = o.egreedy_select(g["step/greedy"][s], g["step/random_actions"][s], g["step/coin"][s], np.float32(eps))
        assert np.array_equal(acts, g["step/acts"][s])
        next_obs, rew, term, trunc = g["step/next_obs"][s], g["step/rewards"][s], g["step/terminals"][s], g["step/truncations"][s]
        buf.store(next_obs if s == 0 else raw, acts, rew, term, next_obs)        # (s == 0: the buf_obs alias, see test_dqn_agent_loop)
        per.store()
        if cur > c["start_training"] and cur % int(c["training_frequency"]) == 0:
            assert int(g[f"phase{phase}/at_step"]) == s and beta == float(g[f"phase{phase}/per/beta"])
            steps, weights = per.sample(beta, g[f"phase{phase}/per/uniforms"])
            assert np.array_equal(steps, g[f"phase{phase}/per/step_choices"]), f"phase {phase}: the transitions the trees pick"
            # (1e-6, not 1e-12: under the NumPy >= 2 of this image the reference's `priority ** alpha` of a float32 |TD error| stays
            #  float32 -- under the NumPy < 2 it pins, and here, it is float64; oracle/xrl_oracle.py: PerBufferOracle)
            assert_close(weights, g[f"phase{phase}/per/weights"], 1e-6, f"phase {phase}: importance weights")
            env_c = np.arange(n).repeat(B // n)
            info, grads = o.dqn_forward_backward(sd, buf.sample_at(env_c, steps.flatten()), dict(gamma=c["gamma"]))
            for name, rg in sub(g, f"phase{phase}/grad0").items():
                assert_close(grads[name], rg, 1e-5, f"phase {phase}: gradient {name}")
            td = np.abs(info["targetQ"] - info["predictQ"])                        # perdqn_learner.py:48
            assert_close(td, g[f"phase{phase}/per/td_error"], 1e-5, f"phase {phase}: |TD error|", scale=max(1.0, float(td.max())))
            per.update_priorities(steps, g[f"phase{phase}/per/td_error"].astype(np.float32))
            opt.step(grads)
            updates += 1
            if updates % int(c["sync_frequency"]) == 0:
                o.dqn_copy_target(sd)
            beta += (1 - beta0) / S
            assert beta == float(g[f"phase{phase}/per_beta_after"])
            phase += 1
        raw = np.where((term | trunc)[:, None], g["step/reset_obs"][s], next_obs)
        cur += n
        if eps > c["end_greedy"]:
            eps -= delta
        assert eps == g["step/eps_after"][s]
    assert phase == int(g["n_phases"])
    leaves = per.sum[:, per.cap:per.cap + per.n_size]
    assert_close(leaves, g["final_priorities"], 1e-6, "priorities in the sum trees' leaves")
    assert_close(per.max_priority, g["final_max_priority"], 1e-6, "running maxima of the priorities")
