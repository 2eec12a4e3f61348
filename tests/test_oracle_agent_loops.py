"""CPU: the ORACLE's restatement of the agent loops (oracle/xrl_oracle.py) pinned to runs of the REFERENCE's own agents
(tests/golden/agent_*.npz, oracle/make_golden_agents.py: the unmodified PPO_Agent.train / DQN_Agent.train / QMIX_Agents.train
recorded through their callback hooks).  Inputs taken from the fixture: the simulators' outputs, the reference's random decisions
(sampled actions, exploration coins, random actions, sample indices).  Everything the loop COMPUTES -- running statistics,
normalised observations, values / log-probs, processed rewards, path closing, GAE, epsilon, greedy actions, update triggers, ring
positions, the parameters after every update phase -- is recomputed by the oracle and compared with the reference's.  The device
loops are compared with the same fixtures in tests/test_gpu_agent_replay.py."""
import numpy as np
import pytest

from conftest import load_golden, sub, assert_close


def test_ppo_agent_loop(oracle):
    """ppo_agent.py:111-181 + core/on_policy.py:182-205: per vector step obs_rms.update(raw obs) -> normalise -> act -> env ->
    store (normalised obs, action, processed reward, value, TERMINATED flag, old_logp); buffer full: V(next_obs) under the CURRENT
    statistics closes every path (0 for terminated envs), the update phase runs, the buffer is cleared; after that the return
    tracker / ret_rms / per-env path closing of finished episodes (no-ops on the emptied buffer when both coincide)."""
    o = oracle
    g = load_golden("agent_ppo")
    c = dict(zip(g["cfg_names"].tolist(), g["cfg"].tolist()))
    n, T, E, MB = (int(c[k]) for k in ("n_envs", "horizon_size", "n_epochs", "n_minibatch"))
    S = g["step/acts"].shape[0]
    sd = {k: v.copy() for k, v in sub(g, "init").items()}
    opt = o.AdamOracle(sd, lr=c["learning_rate"], eps=1e-5, total_iters=int(c["total_iters"]))
    ucfg = dict(vf_coef=c["vf_coef"], ent_coef=c["ent_coef"], clip_range=c["clip_range"], use_grad_clip=True, grad_clip_norm=c["grad_clip_norm"])
    obs_rms, ret_rms = o.RunningMeanStdOracle((4,)), o.RunningMeanStdOracle(())
    returns = np.zeros(n, np.float32)
    buf = o.OnPolicyBufferOracle((4,), (), n, T, gamma=c["gamma"], gae_lam=c["gae_lambda"])
    raw = g["raw_obs0"].copy()
    phase = 0
    for s in range(S):
        obs_rms.update(raw)
        obs_n = o.process_observation(raw, obs_rms, c["obsnorm_range"]).astype(np.float32)
        assert_close(obs_n, g["step/obs"][s], 1e-6, f"step {s}: normalised obs")
        logits, value = o.actor_critic_forward(sd, obs_n)
        probs = np.exp(o.log_softmax(logits))
        assert_close(probs, g["step/probs"][s], 1e-5, f"step {s}: action probabilities")
        acts = g["step/acts"][s]                                                    # fixed input: what the reference sampled
        # (the uniforms the GPU replay supplies reproduce these actions through the inverse CDF)
        cdf = np.cumsum(g["step/probs"][s].astype(np.float64), -1)
        u = (cdf[np.arange(n), acts] - 0.5 * g["step/probs"][s][np.arange(n), acts]).astype(np.float32)
        assert np.array_equal(o.categorical_sample_icdf(logits, u), acts)
        logp = o.log_softmax(logits)[np.arange(n), acts]
        assert_close(value, g["step/vals"][s], 1e-5, f"step {s}: values")
        assert_close(logp, g["step/logp"][s], 1e-5, f"step {s}: log-probs")
        next_obs, rew, term, trunc = g["step/next_obs"][s], g["step/rewards"][s], g["step/terminals"][s], g["step/truncations"][s]
        buf.store(obs_n, acts, o.process_reward(rew, ret_rms, c["rewnorm_range"]), value, term, {"old_logp": logp})
        if buf.full:
            vals = o.actor_critic_forward(sd, o.process_observation(next_obs, obs_rms, c["obsnorm_range"]).astype(np.float32))[1]
            for i in range(n):
                buf.finish_path(0.0 if term[i] else vals[i], i)
            ref = sub(g, f"phase{phase}/buffer")
            assert np.array_equal(buf.actions, ref["actions"]) and np.array_equal(buf.terminals > 0, ref["terminals"] > 0)
            for k in ("observations", "rewards", "values", "returns"):
                assert_close(getattr(buf, k), ref[k], 1e-5, f"phase {phase}: buffer {k}")
            assert_close(buf.old_logp, ref["old_logp"], 1e-5, f"phase {phase}: buffer old_logp")
            assert_close(buf.advantages, ref["advantages"], 1e-5, f"phase {phase}: buffer advantages", scale=float(np.abs(ref["returns"]).max()))
            idx = g[f"phase{phase}/indices"]
            assert idx.shape == (E * MB, n * T // MB)
            for e in range(E):                                                      # each epoch's minibatches partition the buffer
                assert np.array_equal(np.sort(idx[e * MB:(e + 1) * MB].ravel()), np.arange(n * T))
            for k in range(E * MB):
                b = buf.sample(idx[k])
                info, grads = o.ppo_update(sd, opt, dict(obs=b["obs"], actions=b["actions"], returns=b["returns"], advantages=b["advantages"],
                                                         old_logp=b["aux_batch"]["old_logp"]), ucfg)
                ref_g = sub(g, f"phase{phase}/grad{k}")
                for name, rg in ref_g.items():
                    assert_close(info["clipped_grads"][name], rg, 1e-5, f"phase {phase} update {k}: clipped gradient {name}")
            ri = sub(g, f"phase{phase}/info")
            assert_close(info["a_loss"], ri["actor_loss"], 1e-5, "actor_loss", scale=1.0)
            assert_close(info["c_loss"], ri["critic_loss"], 1e-5, "critic_loss")
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
                if term[i]:
                    buf.finish_path(0.0, i)
                else:
                    vals = o.actor_critic_forward(sd, o.process_observation(next_obs, obs_rms, c["obsnorm_range"]).astype(np.float32))[1]
                    buf.finish_path(vals[i], i)
                raw[i] = g["step/reset_obs"][s][i]
        assert_close(returns, g["step/returns_track"][s], 1e-5, f"step {s}: return tracker", scale=max(1.0, float(np.abs(g["step/returns_track"][s]).max())))
        assert_close(obs_rms.mean, g["step/obs_rms/mean"][s], 1e-5, "obs_rms.mean", scale=float(np.sqrt(g["step/obs_rms/var"][s]).max()))
        assert_close(obs_rms.var, g["step/obs_rms/var"][s], 1e-5, "obs_rms.var")
        assert_close(ret_rms.var, g["step/ret_rms/var"][s], 1e-5, "ret_rms.var")
        assert_close(obs_rms.count, g["step/obs_rms/count"][s], 1e-12, "obs_rms.count")
        assert_close(ret_rms.count, g["step/ret_rms/count"][s], 1e-12, "ret_rms.count")
    assert phase == S // T == 3
