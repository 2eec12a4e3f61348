"""Scratch diagnostics for tests/test_gpu_headline.py (second-rollout bootstrap values)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import xrl_oracle as oracle
from test_gpu_headline import c2_config, npy
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 256
torch.manual_seed(0)
env = DeviceCartPoleVecEnv(n, seed=3)
agent = PPO_Agent(c2_config(n, T, use_persistent_rollout=(os.environ.get("PERSIST", "1") == "1")), env)
sd = {k: npy(v) for k, v in agent.model.state_dict().items()}
st = oracle.CartPoleOracle(oracle.cartpole_reset_state(env.seed, np.arange(n), 0))
episodes = np.zeros(n, np.int64)
obs_rms = oracle.RunningMeanStdOracle((4,))
raw_obs = st.state.astype(np.float32)
for it in range(2):
    agent.rollout(); torch.cuda.synchronize()
    f = {k: npy(v) for k, v in agent.memory.soa.fields.items()}
    sd_dev = {k: npy(v) for k, v in agent.model.state_dict().items()}
    worst = 0.0
    for t in range(T):
        prev_mean, prev_var = obs_rms.mean.copy(), obs_rms.var.copy()
        obs_rms.update(raw_obs)
        obs_n = oracle.process_observation(raw_obs, obs_rms).astype(np.float32)
        acts = f["actions"][t].astype(int)
        next_obs, rew, term, trunc = st.step(acts)
        boot = oracle.actor_critic_forward(sd_dev, oracle.process_observation(next_obs, obs_rms).astype(np.float32))[1]
        err = np.abs(f["bootv"][t] - boot)
        done = term | trunc
        if it == 1 and t < 3 or err.max() > 1e-4 and worst < 1e-4:
            class R: pass
            r = R(); r.mean, r.var = prev_mean, prev_var; r.std = np.sqrt(prev_var)
            boot_prev_stats = oracle.actor_critic_forward(sd_dev, np.clip((next_obs - prev_mean) / (np.sqrt(prev_var) + 1e-8), -5, 5).astype(np.float32))[1]
            print(f"it={it} t={t} max|bootv-oracle|={err.max():.3e} at env {err.argmax()} done={np.flatnonzero(done)}")
            print("   dev   ", f["bootv"][t][:8]); print("   oracle", boot[:8]); print("   oracle(prev stats)", boot_prev_stats[:8])
            print("   obs err", np.abs(f["observations"][t] - obs_n).max())
        worst = max(worst, err.max())
        if done.any():
            idx = np.flatnonzero(done); episodes[idx] += 1
            st.state[idx] = oracle.cartpole_reset_state(env.seed, idx, episodes[idx]); st.steps[idx] = 0
        raw_obs = np.where(done[:, None], st.state.astype(np.float32), next_obs)
    print(f"rollout {it}: worst bootv error over ALL (t, env) = {worst:.3e}")
    agent.update(); torch.cuda.synchronize()
