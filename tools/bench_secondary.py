"""MEASUREMENT INFRASTRUCTURE -- the secondary lines of bench.py: the other two workloads the north-star's target sentence
names besides BASELINE configs[1] -- PPO-Clip CartPole-v1 with 16 parallel envs, and QMIX on the SMAC-3m shape (64 envs per
GPU = configs[4] / 8, feed-forward agents and the yaml-default recurrent agents) -- each with its own roofline object.
Every function returns a dict; bench.py attaches them under "secondary".  Timed regions are bracketed by
torch.cuda.synchronize(); kernel / graph times come from HIP events on the launch stream."""
import time
from argparse import Namespace

import torch

PEAK_FP32_MFMA_TFLOPS = 157.3


def _pmc_bytes(which, kernel):
    """HBM bytes per launch of `kernel` from the latest committed PMC summary of the QMIX loops (profiles/r*_qmix_<which>_pmc.json,
    tools/collect_pmc_qmix.sh) or of the C4 loop (which = "c4": profiles/r*_c4_pmc.json, tools/collect_pmc_c4.sh); FETCH_SIZE
    doubled per MI355X_MICROARCH.md -- copied from that builder-run pass, not measured in this run; (None, None) if the file is
    not there."""
    import glob, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r*_%s_pmc.json" % (which if which == "c4" else "qmix_" + which))))
    try:
        for name, v in json.load(open(paths[-1]))["kernels"].items():
            if name.startswith(kernel):
                return int(v["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(paths[-1]) + " (committed rocprofv3 --pmc pass, not measured in this run)"
    except Exception:
        pass
    return None, None


def _pmc_step_bytes(pattern, per_kernel):
    """HBM bytes of ALL kernels of a profiled loop per launch of `per_kernel` (one per update / vector step), from the latest committed
    PMC summary matching `pattern`; (None, None) if missing.  Copied from that builder-run pass, not measured in this run."""
    import glob, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", pattern)))
    try:
        ks = json.load(open(paths[-1]))["kernels"]
        n = [v["launches"] for k, v in ks.items() if k.startswith(per_kernel)][0]
        tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks.values())
        return int(tot / n), ("profiles/" + os.path.basename(paths[-1]) + " (committed rocprofv3 --pmc passes of tools/profile_dqn_c3.py: every kernel "
                              "of the loop -- acting, store, update -- per launch of " + per_kernel + "; not measured in this run)")
    except Exception:
        return None, None


def _c4_rocprof_pair_us():
    """Sum of the average durations of xrl::ppo_wide_kernel and xrl::wide_dw1_kernel in the latest committed rocprofv3 --kernel-trace --stats
    table of the C4 loop (profiles/r*_c4_kernel_stats.csv, tools/collect_pmc_c4.sh), in us; (None, None) if it is not there."""
    import csv, glob, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r*_c4_kernel_stats.csv")))
    try:
        tot = 0.0
        with open(paths[-1]) as f:
            for row in csv.DictReader(f):
                if "xrl::ppo_wide_kernel" in row["Name"] or "xrl::wide_dw1_kernel" in row["Name"]:
                    tot += float(row["AverageNs"]) / 1e3
        return (tot, "profiles/" + os.path.basename(paths[-1])) if tot > 0 else (None, None)
    except Exception:
        return None, None


def _settle():
    """Before a timed window: destroy the graphs / buffers of agents an EARLIER line built now (Python's cycle collector would otherwise
    do it somewhere inside the window -- hipGraphDestroy / hipFree of a whole agent stalls the device for ~40 ms: 72 k instead of 335 k on
    the C3 line in round 4, 23 instead of 9 ms per C4 update in profiles/r05_e_bench.json), then wait for the device."""
    import gc
    gc.collect()
    torch.cuda.synchronize()


def _events_us(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def ppo_small(make_config, kernel_rooflines, n_envs=16, horizon=256, steps=20, warmup=3, ref=None):
    """PPO-Clip CartPole-v1 with 16 parallel envs (the north-star's smallest size): same engine, same graphs."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    torch.manual_seed(1)
    agent = PPO_Agent(make_config(n_envs, horizon, 1, 0), DeviceCartPoleVecEnv(n_envs, seed=1))
    for _ in range(warmup):
        agent.rollout(); agent.update()
    _settle()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.rollout()
        info = agent.update()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "PPO-Clip CartPole-v1, %d envs x horizon %d, 8 epochs x 8 minibatches of %d" % (n_envs, horizon, n_envs * horizon // 8),
           "value": round(n_envs * horizon * steps / dt, 1), "unit": "env-steps/s", "ms_per_step": round(dt / steps * 1e3, 4),
           "steps": steps, "warmup": warmup}
    r1, r2 = kernel_rooflines(agent)
    out["roofline"] = r2 if r2 is not None else r1
    out["roofline_rollout_kernel"] = r1
    if ref:
        out["cpu_baseline"] = ref
    return out


def ppo_acrobot(make_config, n_envs=256, horizon=256, steps=10, warmup=3):
    """PPO-Clip on the device Acrobot-v1 (configs/ppo/classic_control/Acrobot-v1.yaml's network 6-128-{128-3, 128-1} and
    hyper-parameters) at the headline's sizes: a NON-CartPole member of the shared-trunk family -- the any-(D, A) instances of
    ppo_trunk_kernel on 64-row tiles and the general captured rollout (one launch group per vector step; only the (4, 2) class has the
    one-launch rollout kernel) -- so the gap to the specialised class is on record (review item 6 of round 5)."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceAcrobotVecEnv
    torch.manual_seed(1)
    agent = PPO_Agent(make_config(n_envs, horizon, 1, 0), DeviceAcrobotVecEnv(n_envs, seed=1))
    for _ in range(warmup):
        agent.rollout(); agent.update()
    _settle()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.rollout()
        agent.update()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(3):
        agent.rollout()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(3):
        agent.update()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    lr = agent.learner
    return {"workload": "PPO-Clip Acrobot-v1 (device env), %d envs x horizon %d, 8 epochs x 8 minibatches of %d, network 6-128-{128-3, 128-1}"
                        % (n_envs, horizon, n_envs * horizon // 8),
            "value": round(n_envs * horizon * steps / dt, 1), "unit": "env-steps/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "steps": steps, "warmup": warmup, "rollout_ms": round((t2 - t1) / 3 * 1e3, 4), "update_ms": round((t3 - t2) / 3 * 1e3, 4),
            "update_kernel": ("ppo_trunk_bx_kernel<leaky_relu, any (D <= 8, A <= 4)>: 64-row tiles, the 128-wide products as exact 3-way bf16 splits"
                              if getattr(lr, "frag16", None) is not None else
                              "ppo_trunk_kernel<leaky_relu, categorical, %d rows, any (D, A)>" % (64 if getattr(lr, "pair", False) else 32)),
            "rollout_path": ("captured launches per vector step (general path): statistics / bookkeeping, acting pass + sampling as one launch "
                             "(xrl_trunk_forward16), the env's step" if agent._trunk_forward() is not None else
                             "captured launches per vector step (general path)")}


def _qmix_cfg(n, rnn):
    c = dict(q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32, activation="relu", seed=1, parallels=n,
             running_steps=10 ** 7, batch_size=32, learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0,
             end_greedy=0.05, decay_step_greedy=50000, sync_frequency=200, training_frequency=1, n_epochs=8,
             use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True, use_rnn=rnn,
             distributed_training=False, device="cuda", model_dir="/tmp/xrl_bench_models")
    if rnn:   # configs/qmix/sc2/3m.yaml defaults; rnn_backprop_agents False = the reference's behaviour (agents detached)
        c.update(fc_hidden_sizes=[64], recurrent_hidden_size=64, buffer_size=5000, start_training=1000, rnn_backprop_agents=False,
                 episode_length=60)
    else:
        c.update(representation_hidden_size=[64], buffer_size=n * 78, start_training=640)
    return Namespace(**c)


# algorithmic fp32 flops per update: per-agent Q network forward, mixer hyper-networks forward
_FF_AGENT_FWD = 2.0 * (30 * 64 + 64 * 64 + 64 * 9)                        # 13 184 flop per agent row
_RNN_AGENT_FWD = 2.0 * (30 * 64 + 64 * 192 + 64 * 192 + 64 * 64 + 64 * 9)  # fc + W_ih + W_hh + Q head per agent row and step
_MIXER_FWD = 20800.0                                                      # per mixer row (eval); eval fwd+bwd + target fwd = 4x


def qmix_3m(rnn, n=64, steps=None, ref=None):
    """QMIX on the SMAC-3m shape, 64 envs (one GPU's share of BASELINE configs[4]): env-steps/s of the whole agent loop
    (synthetic provider on the device, acting, replay store, 8 updates per vector step / per 64 episodes) and the update
    phase as one captured graph.  Roofline of the update: algorithmic flops of one update / its time (launch-bound)."""
    from xuance_amd.agents import QMIX_Agents
    from xuance_amd.envs import SyntheticSMACVecEnv
    torch.manual_seed(0)
    agent = QMIX_Agents(_qmix_cfg(n, rnn), SyntheticSMACVecEnv(n, seed=3))
    steps = steps or (180 if rnn else 200)
    agent.train(60 if rnn else 20)
    _settle()
    s0, t0 = agent.current_step, time.perf_counter()
    while True:                                     # at least half a second of loop (a 13 ms window does not carry three digits)
        agent.train(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= 0.5:
            break
    env_steps = agent.current_step - s0
    lr = agent.learner
    graph_us = _events_us(lr._buf_graph.launch, 200)
    upd_us = graph_us / 8
    B, N, T = 32, 3, 60
    if rnn:
        flops = (T + 1) * B * N * _RNN_AGENT_FWD * 2 + T * B * _MIXER_FWD * 4     # eval + target agents (detached), mixer f+b + target
        what = "recurrent agents (3m.yaml: fc 64 + GRU 64, 60-step episodes, batch 32 episodes = 1 920 mixer rows, agents detached as in the reference)"
    else:
        flops = B * N * _FF_AGENT_FWD * 5 + B * _MIXER_FWD * 4                    # eval f+b (3x), eval(next), target; mixer as above
        what = "feed-forward agents (Basic_MLP 64 + Q 64-9, batch 32 transitions)"
    tf = flops / upd_us / 1e6
    traffic, traffic_source = _pmc_bytes("gru" if rnn else "ff", "xrl::gru_forward_kernel" if rnn else "xrl::qmix_fused_kernel")
    out = {"workload": "QMIX SMAC-3m shape, %d envs x 3 agents, obs 30 / state 48 / 9 masked actions, %s, 8 updates per %s"
                       % (n, what, "%d episodes" % n if rnn else "vector step"),
           "value": round(env_steps / dt, 1), "unit": "env-steps/s", "update_us": round(upd_us, 2),
           "env_steps_timed": int(env_steps), "seconds": round(dt, 3),
           "roofline": {"bound": "mfma", "kernel": ("update graph (xrl::gemm_f32_kernel launches + xrl::gru_*_kernel + xrl::qmix_prefetch_kernel + "
                                                    "xrl::reduce_adam_kernel)") if rnn else
                                                   "update graph (draw+gather, xrl::qmix_fused_kernel, xrl::reduce_adam_kernel)",
                        "achieved": round(tf, 4), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 5),
                        "traffic": traffic, "traffic_source": traffic_source,
                        "traffic_kernel": "xrl::gru_forward_kernel" if rnn else "xrl::qmix_fused_kernel",
                        "avg_launch_us": round(upd_us, 2), "algorithmic_flops_per_launch": flops,
                        "note": ("one 'launch' = one whole update (a graph of ~9 kernels); launch-latency-bound at batch 32, see DESIGN.md section 3")
                                if rnn else "one 'launch' = one whole update = 3 kernels; the fused kernel is VALU fp32 on 32 workgroups (one "
                                            "transition each), latency-bound: DESIGN.md section 3"}}
    if ref:
        out["cpu_baseline"] = ref
    return out


def ppo_c4(steps=5, warmup=2, ref=None):
    """BASELINE configs[3] shapes on one GPU: PPO, Gaussian policy 17-256-256-6 + critic 17-256-256-1 (configs/ppo/mujoco.yaml),
    128 envs x horizon 256, 16 epochs x 8 minibatches of 4 096, MuJoCo-shaped synthetic provider on the device.  Update: ONE
    launch per minibatch for forward + loss + backward (xrl::ppo_wide_kernel, csrc/ppo_wide.hip) + the optimiser launch;
    rollout: two launches per vector step (acting incl. statistics / bookkeeping, provider).  Roofline of the dominant
    kernel: algorithmic flops of one minibatch (849 408 flop per sample) / its HIP-event-timed launch."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticMujocoVecEnv
    n, T = 128, 256
    cfg = Namespace(agent="PPO", representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                    critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", seed=1, parallels=n,
                    running_steps=10 ** 9, horizon_size=T, n_epochs=16, n_minibatch=8, learning_rate=4e-4, vf_coef=0.25,
                    ent_coef=0.0, clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
                    grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_bench_models", use_hip_graph=True)
    torch.manual_seed(0)
    agent = PPO_Agent(cfg, SyntheticMujocoVecEnv(n, seed=4))
    for _ in range(warmup):
        agent.rollout(); agent.update()
    _settle()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.rollout()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(steps):
        agent.update()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    us_mb = (t2 - t1) / steps / 128 * 1e6
    flops = 849408.0 * 4096
    lr = agent.learner
    wide = getattr(lr, "_wide", None)
    if wide is not None:                                # the one-launch minibatch kernel, timed by itself on staged rows
        bs, P = agent.batch_size, agent.model.params.P
        st = {k: v[3 * bs:4 * bs] for k, v in lr._wstage.items()}
        fn = lambda: wide.launch(bs, st["observations"], st["actions"], st["returns"], st["advantages"], st["aux_old_logp"],
                                 lr.fslabs, P, lr.fpartials, lr.clip_range, lr.vf_coef, lr.ent_coef, stats=lr.stats[3])
        for _ in range(10):
            fn()
        us_k = _events_us(fn, 200)
        # the minibatch's matrix work is TWO launches since round 4 (xrl::ppo_wide_kernel + xrl::wide_dw1_kernel: the middle layers' weight
        # gradient over all rows); HIP events bracket the pair.  The committed rocprofv3 kernel table of the same loop is reported next to
        # it, and `frac` is taken from the LARGER of the two times (the profiler adds a few us per kernel; the review computes from it)
        prof_us, prof_src = _c4_rocprof_pair_us()
        if prof_us is not None and prof_us > us_k:
            us_events, us_k = us_k, prof_us
        else:
            us_events = us_k
        kname, note = "xrl::ppo_wide_kernel + xrl::wide_dw1_kernel", (
            "forward + Gaussian PPO-clip loss + backward of one 4 096-row minibatch: two launches (the middle layers' weight gradient over all rows is the "
            "second); HIP events around the pair: %.1f us; sum of the two kernels' average durations in %s: %s us; frac uses the larger.  With the optimiser "
            "launch (xrl::reduce_adam_kernel) a minibatch takes %.1f us = %.3f of peak"
            % (us_events, prof_src, "%.1f" % prof_us if prof_us is not None else "n/a", us_mb, flops / us_mb / 1e6 / PEAK_FP32_MFMA_TFLOPS))
    else:
        us_k, kname, note = us_mb, "minibatch update (xrl::gemm_f32_kernel launches + xrl::ppo_loss_kernel + xrl::reduce_adam_kernel)", \
            "one 'launch' = one whole minibatch update (layered path)"
    tf = flops / us_k / 1e6
    traffic, traffic_src = _pmc_bytes("c4", "xrl::ppo_wide_kernel") if wide is not None else (None, None)   # (of the first of the two launches)
    out = {"workload": "PPO, HalfCheetah shapes (obs 17, Box(6), Gaussian 17-256-256-6 + critic 17-256-256-1), %d envs x horizon %d, "
                       "16 epochs x 8 minibatches of 4096 (BASELINE configs[3], per GPU)" % (n, T),
           "value": round(n * T * steps / (t2 - t0), 1), "unit": "env-steps/s", "ms_per_step": round((t2 - t0) / steps * 1e3, 3),
           "rollout_ms": round((t1 - t0) / steps * 1e3, 3), "update_ms": round((t2 - t1) / steps * 1e3, 3),
           "update_us_per_minibatch": round(us_mb, 1),
           "roofline": {"bound": "mfma", "kernel": kname, "achieved": round(tf, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_us": round(us_k, 1), "algorithmic_flops_per_launch": flops, "note": note}}
    if ref:
        out["cpu_baseline"] = ref
    return out


def ppo_atari(steps=3, warmup=2, ref=None):
    """configs/ppo/atari.yaml shapes: PPO on 84x84x4 uint8 frame stacks, AC_CNN_Atari (32/64/64 conv + 512 dense) with a categorical
    head, 8 envs x horizon 128 in a HipOnPolicyBuffer_Atari (uint8), 4 epochs x 4 minibatches of 256 frames; synthetic frame
    provider on the device.  The network is 3.36 M parameters (the 6400 -> 512 layer); roofline of a minibatch: its algorithmic
    flops (conv 21.2 MFLOP + dense 6.6 MFLOP per frame and forward pass, x3 for forward + two backward products) / its time."""
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n, T = 8, 128
    cfg = Namespace(agent="PPO", representation="AC_CNN_Atari", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                    fc_hidden_sizes=[512], actor_hidden_size=[], critic_hidden_size=[], activation="relu", seed=1, parallels=n,
                    running_steps=10 ** 7, horizon_size=T, n_epochs=4, n_minibatch=4, learning_rate=2.5e-4, vf_coef=0.25, ent_coef=0.01,
                    clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True, grad_clip_norm=0.5,
                    use_obsnorm=False, use_rewnorm=False, obsnorm_range=5, rewnorm_range=5, distributed_training=False, device="cuda",
                    model_dir="/tmp/xrl_bench_models", use_hip_graph=True)
    torch.manual_seed(0)
    agent = PPO_Agent(cfg, SyntheticAtariVecEnv(n, seed=5))
    for _ in range(warmup):
        agent.rollout(); agent.update()
    import gc
    gc.collect()                                    # (see dqn_c3)
    _settle()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.rollout()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(steps):
        agent.update()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    us_mb = (t2 - t1) / steps / 16 * 1e6
    flops = 3 * (21.2e6 + 2 * 6400 * 512 + 2 * 512 * 5) * 256
    tf = flops / us_mb / 1e6
    out = {"workload": "PPO, Atari shapes (84x84x4 uint8 frames, AC_CNN_Atari 32/64/64 + 512, 4 actions; configs/ppo/atari.yaml), %d envs x "
                        "horizon %d, 4 epochs x 4 minibatches of 256 frames, uint8 rollout buffer" % (n, T),
            "value": round(n * T * steps / (t2 - t0), 1), "unit": "env-steps/s", "ms_per_step": round((t2 - t0) / steps * 1e3, 3),
            "rollout_ms": round((t1 - t0) / steps * 1e3, 3), "update_ms": round((t2 - t1) / steps * 1e3, 3),
            "update_us_per_minibatch": round(us_mb, 1),
            "roofline": {"bound": "mfma", "kernel": "minibatch update (implicit-GEMM convolutions xrl::conv_mfma_kernel / conv_dw_*, dense xrl::gemm_f32_kernel "
                                                    "launches, xrl::ppo_loss_kernel, xrl::reduce_adam_kernel over 3.36 M parameters)",
                         "achieved": round(tf, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": None, "avg_launch_us": round(us_mb, 1), "algorithmic_flops_per_launch": flops,
                         "note": "one 'launch' = one whole minibatch update (layered path over the convolution stack); the rollout is one captured graph of 128 vector steps of 8 envs (~15 launches per step: launch-bound at 8 envs)"}}
    if ref:
        out["cpu_baseline"] = ref
    return out


def dqn_c3(steps=200, ref=None, buffer_size=499968, start_training=10000):
    """BASELINE configs[2] AT ITS CONFIGURATION (configs/dqn/atari.yaml with 64 envs): DQN, 64 envs x 84x84x4 uint8 frames (synthetic
    frame provider on the device), CNN 32/64/64 + 512, uint8 replay ring of 499 968 transitions = 7 812 slots x 64 envs x 2 x 28 224 B
    = 28.2 GB in HBM (memory_tools.py:105,354: buffer_size // n_envs slots per env), batch 32, start_training 10 000, one update per
    vector step (training_frequency = n_envs).  The ring is FILLED before the timed window (HipOffPolicyBuffer.fill_synthetic: the
    loop would need 7 812 vector steps to do it) and the loop is past start_training, so every sampled row comes from anywhere
    in the 28 GB.  Roofline of the update graph: 2.7 GFLOP (3x eval + 1x target forward-equivalents of the 21.2 MFLOP
    network at batch 32) / its time; `replay_gather`: the draw + gather launch alone (56 448 B per sampled transition)."""
    from xuance_amd.agents import DQN_Agent
    from xuance_amd.envs import SyntheticAtariVecEnv
    n = 64
    cfg = Namespace(env_name="Atari", representation="Basic_CNN", kernels=[8, 4, 3], strides=[4, 2, 1], filters=[32, 64, 64],
                    q_hidden_size=[512], activation="relu", seed=1, parallels=n, running_steps=10 ** 7,
                    buffer_size=int(buffer_size), batch_size=32, learning_rate=1e-4, gamma=0.99, start_greedy=0.5, end_greedy=0.05,
                    decay_step_greedy=10 ** 6, sync_frequency=500, training_frequency=n, start_training=int(start_training),
                    use_grad_clip=False, grad_clip_norm=0.5, use_obsnorm=False, use_rewnorm=False,
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_bench_models")
    torch.manual_seed(0)
    agent = DQN_Agent(cfg, SyntheticAtariVecEnv(n, seed=2))
    agent.memory.fill_synthetic(seed=4)
    agent.train(start_training // n + 16)            # past start_training (`current_step > start_training`, off_policy.py:228) + warm-up updates
    assert agent.learner.iterations > 0 and agent.memory.size == agent.memory.n_size == buffer_size // n
    _settle()
    n_done, t0 = 0, time.perf_counter()
    while True:                                     # at least half a second of loop
        agent.train(steps)
        n_done += steps
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= 0.5:
            break
    steps = n_done
    lr = agent.learner
    lr.update_from_buffer(agent.memory, 1, seed=1)
    graph_us = _events_us(lr._buf_graph.launch, 20)
    # the draw + gather launch alone, rows from anywhere in the full ring
    mem, M = agent.memory, 32
    dst = {"observations": lr.X[:M], "next_observations": lr.X[M:2 * M], "actions": lr._act, "rewards": lr._rew, "terminals": lr._ter}
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    gather_us = _events_us(lambda: mem.draw_into(lr._idx, dst, 7, 0, ctr), 200)
    gather_bytes = M * (2 * 84 * 84 * 4 + 12)
    flops = 4 * 21.2e6 * 32
    tf = flops / graph_us / 1e6
    traffic, traffic_src = _pmc_step_bytes("r*_dqn_c3_pmc.json", "xrl::dqn_tail_td_kernel")
    out = {"workload": "DQN, Atari shapes (84x84x4 uint8 frames, CNN 32/64/64 + 512, 4 actions), %d envs, uint8 replay ring, batch 32, "
                       "one update per vector step (BASELINE configs[2] at its configuration: 499 968-transition ring = 28.2 GB, start_training 10 000)" % n,
           "value": round(n * steps / dt, 1), "unit": "env-steps/s", "vector_step_us": round(dt / steps * 1e6, 1),
           "update_us": round(graph_us, 1),
           "replay": {"slots_per_env": mem.n_size, "transitions": mem.n_size * n, "ring_GB": round(mem.n_size * n * 2 * 84 * 84 * 4 / 1e9, 2),
                      "start_training": int(start_training), "filled": "HipOffPolicyBuffer.fill_synthetic before the timed window"},
           "replay_gather": {"us": round(gather_us, 2), "algorithmic_bytes": gather_bytes, "GBps": round(gather_bytes / gather_us / 1e3, 1),
                             "note": "xrl_soa_gather_sampled at batch 32: 1.8 MB per launch out of a 28 GB ring -- latency-bound (one 28 KB row per "
                                     "4 workgroups); profiles/r05_scale_gather.json has the kernel at batches where HBM bounds it"},
           "roofline": {"bound": "mfma", "kernel": "update graph (xrl::conv_mfma_kernel / conv_dw_mfma_kernel implicit GEMMs + Q-head launches, eval + target networks, backward, xrl::reduce_adam_kernel)",
                        "achieved": round(tf, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                        "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(graph_us, 1), "algorithmic_flops_per_launch": flops,
                        "note": "one 'launch' = one whole update (a graph); 10 launches at batch 32 (replay gather, 3 convolutions, xrl_dqn_tail_td, 2 input-gradient + 2 weight-gradient convolutions, xrl_reduce_adam): launch- and latency-bound, DESIGN.md section 3 'Round 3: implicit-GEMM convolutions'"}}
    if ref:
        out["cpu_baseline"] = ref
    return out
