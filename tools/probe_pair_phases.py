"""Scratch: where a minibatch launch of the 64-row role-split kernel (ppo_trunk_kernel<.., 64>; and of the 32-row form for comparison) spends its time -- shader-clock
stamps of the last workgroup's phases, the 100 MHz real-time counter at the start / end of every workgroup (launch skew, slowest
workgroup, tail) and the HIP-event time of the launch alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from xuance_amd import ops
from xuance_amd.agents import PPO_Agent
from xuance_amd.envs import DeviceCartPoleVecEnv
n = 256
for mode in ("pair-bx", "pair-bx-actor", "pair", "tile32"):      # ("chain": round 4's variant, tools/csrc/ppo_chain.hip, no longer in the library)
    pair = mode != "tile32"
    cfg = bench.make_config(n, 256, 1, 0); cfg.use_pair_update = pair; cfg.use_split_products = mode.startswith("pair-bx")
    torch.manual_seed(1)
    agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
    agent.rollout(); agent.update(); torch.cuda.synchronize()
    lr, mem, m = agent.learner, agent.memory, agent.model
    f, bs = mem.soa.fields, agent.batch_size
    dbg = torch.zeros(2048, dtype=torch.int64, device="cuda")

    def launch(d):
        ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                                f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                                f_logp=f["aux_old_logp"], idx=agent.idx[3], stats=lr.stats[3], slabs=lr.fslabs, partials=lr.fpartials,
                                diag=None, slab_stride=lr.slab_stride, l0_fold_off=lr.fold[0] if lr.fold else 0, M=bs, n_envs=n, T=256,
                                D=4, A=2, clip_range=0.2, vf_coef=0.25, ent_coef=0.01, dbg=d, frag_image=lr.frag, f_packed=lr.packed,
                                f_rows=lr.rows[3 * bs * 8:4 * bs * 8], pad0=64 if lr.pair else 0,
                                frag16=lr.frag16 if mode.startswith("pair-bx") else None, pad3=2 if mode == "pair-bx-actor" else 0)
    for _ in range(3):
        launch(dbg)
        torch.cuda.synchronize()
    us = bench._event_time_us(lambda: launch(None), 50)
    d = dbg.cpu().numpy()
    if mode == "chain":
        names = ["start", "rows + small parameters", "weight block + h1 in LDS", "branch layer, head, loss, g2 (chain waves)", "dW1 || dH1", "g1 stored", "end"]
        print("chain: alone %.1f us; phases (cycles):" % us, dict(zip(names, (d[:7] - d[0]).tolist())))
    elif pair:
        names = ["start", "rows gathered", "h1", "h2 (fwd MFMA)", "heads+loss+g2", "small grads", "dW1", "dH1+g1", "end"]
        ph = d[:9] - d[0]
        t = d[16:16 + 2 * 256].reshape(256, 2).astype(np.float64) * 10e-3      # us
        t0 = t[:, 0].min()
        print("%s: alone %.1f us; phases (cycles):" % (mode, us), dict(zip(names, ph.tolist())))
        if mode.startswith("pair-bx"):
            print("  split-product kernel, inside 'dH1+g1' (cycles from start): dH1 issued + dW stores queued %d | late barrier %d | n-halves met %d | first-layer sums formed %d | barrier #4 %d"
                  % tuple(int(d[k] - d[0]) for k in (9, 10, 11, 12, 7)))
        w = d[1100:1100 + 128].reshape(8, 16)[:, :8] - d[0]
        print("  waves of the last workgroup (cycles from its start): small grads done | dW1 MFMAs done | dW1 stored | dH1 MFMAs done | past barrier 4 | g1 stored | past barrier 5 | end")
        for i in range(8):
            print("   wave", i, w[i].tolist())
        print("  workgroup start skew us: min %.2f max %.2f; durations us: min %.2f median %.2f max %.2f; last end %.2f"
              % (0.0, (t[:, 0] - t0).max(), (t[:, 1] - t[:, 0]).min(), np.median(t[:, 1] - t[:, 0]), (t[:, 1] - t[:, 0]).max(), (t[:, 1] - t0).max()))
    else:
        print("32-row tiles: alone %.1f us" % us)
