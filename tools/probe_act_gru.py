"""The recurrent agents' acting launch (xrl_marl_act_gru, 3m.yaml shapes: 64 envs x 3 agents = 192 rows, fc 30 -> 64, GRU 64, Q 64 -> 9)
alone: average launch time from a graph of 64 launches, for several rows-per-workgroup settings, and the per-workgroup phase stamps
(xrl_debug_act_gru_stamps).  usage: probe_act_gru.py [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np
import torch
from xuance_amd import ops, _lib
from xuance_amd.agents import QMIX_Agents
from xuance_amd.envs import SyntheticSMACVecEnv

n = 64
cfg = Namespace(q_hidden_size=[64], fc_hidden_sizes=[64], recurrent_hidden_size=64, hidden_dim_mixing_net=32,
                hidden_dim_hyper_net=32, activation="relu", seed=1, parallels=n, running_steps=10 ** 7, buffer_size=5000,
                batch_size=32, learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0, end_greedy=0.05,
                decay_step_greedy=50000, sync_frequency=200, training_frequency=1, start_training=0, n_epochs=8,
                use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True, use_rnn=True,
                rnn_backprop_agents=False, episode_length=60, distributed_training=False, device="cuda", model_dir="/tmp/x")
torch.manual_seed(0)
agent = QMIX_Agents(cfg, SyntheticSMACVecEnv(n, seed=3))
model = agent.model if hasattr(agent, "model") else agent.policy
R = n * 3
dev = "cuda"
obs = torch.randn(R, 30, device=dev)
h = torch.randn(R, 64, device=dev) * 0.1
qo = torch.zeros(R, 9, device=dev)
act, actf = torch.zeros(R, dtype=torch.int32, device=dev), torch.zeros(R, device=dev)
avail = torch.ones(R, 9, device=dev)
eps = torch.full((1,), 0.3, device=dev)
out = []
for variant in (os.environ.get("XRL_QA_VARIANTS", "0,1").split(",")):
    for rpw in ((6, 3, 2, 1) if variant == "0" else (1,)):
        st = ops.MarlActGruState(model, rows_per_wg=rpw, lds_staged=(variant == "0"))
        if st.lds_bytes > 160 * 1024:
            continue
        sel = dict(action=act, action_f=actf, avail=avail, eps_dev=eps, seed=5, step=0, step_dev=None)
        go = lambda: st.launch(obs, R, h, None, qo, select=sel)
        go(); torch.cuda.synchronize()
        g = ops.Graph()
        with g:
            for _ in range(64):
                go()
        for _ in range(3):
            g.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.launch()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 / 64 * 1e6
        n_wg = (R + rpw - 1) // rpw
        stamps = torch.zeros(16 * n_wg, dtype=torch.int64, device=dev)
        _lib.call("xrl_debug_act_gru_stamps", stamps.data_ptr())
        go(); torch.cuda.synchronize()
        _lib.call("xrl_debug_act_gru_stamps", None)
        s = stamps.cpu().numpy().reshape(n_wg, 16)
        ph = np.diff(s[:, 1:8], axis=1)                      # cycles between phase boundaries (100 MHz s_memtime? no: shader clock)
        rt = (s[:, 15] - s[:, 0].min()) * 0.01               # us since the first workgroup started (s_memrealtime: 100 MHz)
        rec = dict(variant=variant, rows_per_wg=rpw, workgroups=n_wg, launch_us=round(us, 2), lds_bytes=st.lds_bytes,
                   phase_cycles_median=dict(zip(["load+stage", "reset+fc", "gates", "cell", "q_head", "store+select"],
                                                [float(x) for x in np.median(ph, axis=0)])),
                   start_spread_us=round(float((s[:, 0].max() - s[:, 0].min()) * 0.01), 2), end_us_max=round(float(rt.max()), 2),
                   end_us_median=round(float(np.median(rt)), 2))
        print(json.dumps(rec)); out.append(rec)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
