"""xrl_act_tail alone (256 envs, 6-128-{128-3, 128-1} heads on a 256-wide level): launch time with and without the env's step, from a
graph of 64 launches; the launches it replaces (heads product, xrl_policy_sample, xrl_classic_step) timed the same way."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd import ops
from xuance_amd.envs import DeviceAcrobotVecEnv, DeviceMountainCarVecEnv

def timed(fn, reps=20, inner=64):
    fn(); torch.cuda.synchronize()
    g = ops.Graph()
    with g:
        for _ in range(inner):
            fn()
    for _ in range(3):
        g.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / inner * 1e6

n, A, K = 256, 3, 128
dev = "cuda"
torch.manual_seed(0)
hb = torch.randn(2 * n, 256, device=dev)
wa, ba, wc, bc = torch.randn(A, K, device=dev) * 0.1, torch.zeros(A, device=dev), torch.randn(1, K, device=dev) * 0.1, torch.zeros(1, device=dev)
heads = torch.zeros(2 * n, A + 1, device=dev)
out = {}
for Env in (DeviceAcrobotVecEnv, DeviceMountainCarVecEnv):
    env = Env(n, seed=1)
    env.reset()
    act, val, logp, bootv = (torch.zeros(n, device=dev) for _ in range(4))
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    sample = dict(heads=None, log_std=None, noise=None, act_out=act, val_out=val, logp_out=logp, env_action=env.action, env_action_f=None,
                  bootv_prev=bootv, n=n, A=A, ld=A + 1, gaussian=0, seed=3, step=0, step_dev=step)
    common = dict(hb=hb, ldh=256, K=K, a_off=0, c_off=128, w_actor=wa, b_actor=ba, w_critic=wc, b_critic=bc, ldw_a=K, ldw_c=K, heads=heads,
                  boot_rows=1, boot_actor=0, act_actor=0)
    r = {}
    r["tail_with_env_us"] = timed(lambda: ops.act_tail(sample=sample, env_kind=env.kind, classic=env._kw(), **common))
    r["tail_without_env_us"] = timed(lambda: ops.act_tail(sample=sample, env_kind=0, **common))
    r["env_step_us"] = timed(lambda: env.step_device())
    r["policy_sample_us"] = timed(lambda: ops.policy_sample(**dict(sample, heads=heads)))
    from xuance_amd import _lib
    st = torch.zeros(8, dtype=torch.int64, device=dev)
    _lib.call("xrl_debug_act_tail_stamps", st.data_ptr())
    ops.act_tail(sample=sample, env_kind=env.kind, classic=env._kw(), **common); torch.cuda.synchronize()
    _lib.call("xrl_debug_act_tail_stamps", None)
    d = st.cpu().numpy()
    r.update(cycles_stage=float(d[1] - d[0]), cycles_heads=float(d[2] - d[1]), cycles_sample=float(d[3] - d[2]), cycles_env=float(d[4] - d[3]))
    out[Env.__name__] = {k: round(v, 2) for k, v in r.items()}
    print(Env.__name__, json.dumps(out[Env.__name__]))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
