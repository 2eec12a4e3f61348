"""Micro-benchmark of the rollout-side launches (run on the GPU box): per-launch time of the fused step kernel in its
full and bootstrap-only forms, from a 200-node hipGraph (so host launch cost is excluded)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
from bench import make_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
agent = PPO_Agent(make_config(n, 256, 1, 0), DeviceCartPoleVecEnv(n, seed=1))
agent.rollout(); torch.cuda.synchronize()


def timed(fn, reps=200, rounds=5):
    g = ops.Graph()
    with g:
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.launch(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


T, D, A = agent.horizon_size, agent.obs_dim, agent.model.action_dim
env, f, pp = agent.envs, agent.memory.soa.fields, agent.pp
common = dict(params=agent.model.params.flat, cache_image=agent.cache_image, frag_image=agent.frag_image, role_split=int(os.environ.get('SPLIT','1')), split_col=128, ret_track=agent.returns, cp_state=env.state, cp_steps=env.steps,
              cp_episodes=env.episodes, cp_score=env.ep_score, cp_stats=env.stats, n=n, D=D, A=A, gaussian=0,
              max_steps=500, use_obsnorm=1, use_rewnorm=1, obs_range=5.0, rew_range=5.0, gamma=0.98, seed=1, env_seed=1,
              step_dev=agent.step_counter)


def full(t=5, boot=True):
    i, o = t & 1, (t + 1) & 1
    ops.rollout_step_cartpole(agent.model.plan, obs_raw_in=pp["obs_raw"][i], obs_raw_out=pp["obs_raw"][o],
        xnext_in=pp["xnext"][i], xnext_out=pp["xnext"][o], obs_stats_in=pp["obs_stats"][i], obs_stats_out=pp["obs_stats"][o],
        obs_count_in=pp["obs_count"][i], obs_count_out=pp["obs_count"][o], ret_stats_in=pp["ret_stats"][i],
        ret_stats_out=pp["ret_stats"][o], ret_count_in=pp["ret_count"][i], ret_count_out=pp["ret_count"][o],
        ended_in=pp["ended"][i], ended_out=pp["ended"][o], ret_final_in=pp["ret_final"][i], ret_final_out=pp["ret_final"][o],
        obs_slot=f["observations"][t], act_slot=f["actions"][t], val_slot=f["values"][t], logp_slot=f["aux_old_logp"][t],
        rew_slot=f["rewards"][t], term_slot=f["terminals"][t], seg_slot=f["seg"][t],
        bootv_prev=f["bootv"][t - 1] if boot else None, last_step=0, boot_only=0, step=t, **common)


def boot_only():
    ops.rollout_step_cartpole(agent.model.plan, xnext_in=pp["xnext"][0], bootv_prev=f["bootv"][T - 1], boot_only=1,
                              last_step=0, step=0, **common)


def empty():
    ops.counter_add(agent.step_counter, 0)


print(f"n_envs={n}")
print("trivial 1-thread kernel      : %.2f us" % timed(empty))
print("fused step (act+boot tiles)  : %.2f us" % timed(full))
print("fused step (act tiles only)  : %.2f us" % timed(lambda: full(boot=False)))
print("fused boot-only (MLP only)   : %.2f us" % timed(boot_only))
X = agent.X
print("unfused 3-layer forward M=2n : %.2f us" % timed(lambda: agent.model.forward(X, 2 * n)))

dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
def full_dbg(t=5):
    i, o = t & 1, (t + 1) & 1
    ops.rollout_step_cartpole(agent.model.plan, obs_raw_in=pp["obs_raw"][i], obs_raw_out=pp["obs_raw"][o],
        xnext_in=pp["xnext"][i], xnext_out=pp["xnext"][o], obs_stats_in=pp["obs_stats"][i], obs_stats_out=pp["obs_stats"][o],
        obs_count_in=pp["obs_count"][i], obs_count_out=pp["obs_count"][o], ret_stats_in=pp["ret_stats"][i],
        ret_stats_out=pp["ret_stats"][o], ret_count_in=pp["ret_count"][i], ret_count_out=pp["ret_count"][o],
        ended_in=pp["ended"][i], ended_out=pp["ended"][o], ret_final_in=pp["ret_final"][i], ret_final_out=pp["ret_final"][o],
        obs_slot=f["observations"][t], act_slot=f["actions"][t], val_slot=f["values"][t], logp_slot=f["aux_old_logp"][t],
        rew_slot=f["rewards"][t], term_slot=f["terminals"][t], seg_slot=f["seg"][t],
        bootv_prev=None, last_step=0, boot_only=0, step=t, dbg=dbg, **common)


nt = (n + 31) // 32
for name, fn, blk in (("boot-only", lambda: ops.rollout_step_cartpole(agent.model.plan, xnext_in=pp["xnext"][0], bootv_prev=f["bootv"][T - 1], boot_only=1, last_step=0, step=0, dbg=dbg, **common), 0),
                      ("role0 act-actor ", full_dbg, 0), ("role1 act-critic", full_dbg, nt)):
    for _ in range(3):
        dbg.zero_(); dbg[63] = blk
        fn(); torch.cuda.synchronize()
    d = dbg.tolist(); k = d[15]
    print(name, "phase cycles:", [d[i + 1] - d[i] for i in range(k - 1)], "total", d[k - 1] - d[0])
    if d[16] and d[17] and d[32] == 0:
        print("   per-wave section (start, end) relative to kernel start:", [(d[16 + 2 * w] - d[0], d[17 + 2 * w] - d[0]) for w in range(8)])
    pr = [] if d[32] == 0 else [x for x in d[16:63] if x]
    if pr:
        print("   in-layer probe deltas (enter, mfma done, epilogue done, barrier done ...):", [pr[i + 1] - pr[i] for i in range(len(pr) - 1)])
