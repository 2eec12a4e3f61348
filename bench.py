#!/usr/bin/env python
"""Benchmark of the hot path: env-steps/sec of PPO-Clip rollout+update on CartPole-v1 (BASELINE.json configs[1]:
256 parallel envs per GPU, horizon 256, 8 epochs x 8 minibatches of 8192, net 4->128->{128->2,128->1}).

  python bench.py --gpus N --steps K --warmup W [--workload c2|c4|qmix3m|qmix3m_gru]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch = one rollout of horizon_size vector steps on n_envs envs
(policy inference, device CartPole physics, SoA store, obs/reward normalisation, GAE) followed by the full update
phase (n_epochs x n_minibatch minibatch updates: gather, fp32-MFMA forward/backward, PPO loss, clip, Adam).
Nothing is skipped inside the timed region.  Weak scaling: every rank owns n_envs envs and its own buffer, gradients are
averaged once per optimiser step.  Rank 0 prints ONE JSON line.

N > 1: before the timed region every usable way of averaging the gradients (inside the optimiser launch through IPC-mapped
peer buffers / RCCL all-reduce captured in the update graph / graphs cut at the all-reduce; xuance_amd/dist.py) is TIMED on
the real workload and the fastest is adopted -- `config.gradient_paths_ms` holds all timings, `config.gradient_average` the one
the timed region used, `config.rccl_world` a one-element all-reduce, `config.rollout_mode` how the rollouts ran.  The N-rank
run also carries `secondary` lines for the two other configurations BASELINE.json sizes for a node: configs[3] (`c4`,
HalfCheetah shapes, 128 envs/GPU) and configs[4] (`qmix3m`, 64 envs/GPU), so that a scaling run yields their curves too
(`--workload` makes either the main line instead).
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_workloads as bw  # noqa: E402

make_config = bw.c2_config          # (tools/bench_secondary.py builds its 16-env agent from the same configuration)


_PMC_SOURCE = [None]
_LIVE_PMC = {}          # kernel name (up to its argument list) -> HBM bytes per launch, measured by live_pmc() in THIS run


def live_pmc(timeout_s=150):
    """HBM traffic of the bench's kernels measured in this run: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE:
    separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes) over a 3-step child of this command, FETCH_SIZE doubled
    (gfx950 tallies a wide coalesced stream at 64 B per 128-B request), WRITE_SIZE raw; fills _LIVE_PMC.  Returns a note (why it did
    not run: no rocprofv3, a pass failed or timed out -- the committed pass of profiles/ is reported then, and says so)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-roofline"]
    acc = {"FETCH_SIZE": defaultdict(lambda: [0.0, 0]), "WRITE_SIZE": defaultdict(lambda: [0.0, 0])}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(d, ctr)
            pr = subprocess.Popen([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "x", "--"] + child, cwd="/tmp",
                                  env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                return "rocprofv3 --pmc %s pass timed out after %d s" % (ctr, timeout_s)
            if rc != 0:
                return "rocprofv3 --pmc %s pass exited with %d" % (ctr, rc)
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get("Counter_Name") == ctr:
                            a = acc[ctr][row["Kernel_Name"]]
                            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k in set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"]):
        fr = acc["FETCH_SIZE"][k][0] / max(acc["FETCH_SIZE"][k][1], 1)
        wr = acc["WRITE_SIZE"][k][0] / max(acc["WRITE_SIZE"][k][1], 1)
        _LIVE_PMC[k.split("(")[0].replace("void ", "")] = int((2 * fr + wr) * 1024)
    return None if _LIVE_PMC else "the passes produced no counter rows"


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel`: measured in this run when live_pmc() ran (the default at N = 1), else from the committed
    rocprofv3 PMC passes of this same command (profiles/, FETCH_SIZE doubled per MI355X_MICROARCH.md) -- `traffic_source` says
    which; None if neither exists."""
    import glob
    for name, v in _LIVE_PMC.items():
        if name.startswith(kernel):
            _PMC_SOURCE[0] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over a 3-step "
                              "child of this command; 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes")
            return v
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ppo_c2_pmc_hbm.json")))      # latest committed pass
    try:
        with open(paths[-1]) as f:
            ks = json.load(f)["kernels"]
        for name, v in ks.items():                       # template instantiations carry their arguments in the name
            if name.startswith(kernel):
                _PMC_SOURCE[0] = "profiles/" + os.path.basename(paths[-1]) + " (committed rocprofv3 --pmc pass of this command, not measured in this run)"
                return int(v["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def reference_cpu_baseline(key, sub=None):
    """The CPU baseline of record: the UNMODIFIED reference timed by oracle/time_reference_cpu.py through its own
    agent.train() (profiles/ref_cpu_baseline.json, cores stated).  The reference tree does not exist on the GPU box, so it
    cannot be re-timed here; the file says where it was measured."""
    try:
        with open(os.path.join(ROOT, "profiles", "ref_cpu_baseline.json")) as f:
            r = json.load(f)
        e = r[key][sub] if sub is not None else r[key]
        return {"value": e["env_steps_per_s"], "unit": "env-steps/s", "cores": r["cores"], "kind": "reference",
                "sample": e["what"] + "; median of %d timed calls after one warm-up call; unmodified reference on device 'cpu', "
                          "torch %s, %d threads; measured on: %s (profiles/ref_cpu_baseline.json, oracle/time_reference_cpu.py)"
                          % (len(e["runs"]), r["torch"], r["threads"], r["host"])}
    except Exception as ex:                                  # noqa: BLE001
        return {"value": None, "kind": "reference", "error": repr(ex)}


def _event_time_us(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def kernel_rooflines(agent):
    """Rooflines of the two kernels that make up the timed region (profiles/*.csv), timed live with HIP events on the
    launch stream: (1) xrl::actor_rollout_kernel + xrl::critic_values_kernel -- the T vector steps of a rollout as ONE launch
    with only the actor on the step chain, then values / bootstrap values of the rollout as one batched launch (or, when the
    network is not of that class, xrl::rollout_step_cartpole_kernel, T + 1 launches) -- and
    (2) xrl::ppo_trunk_kernel -- one launch per minibatch, (64-row tile, role) workgroups -- measured inside the real minibatch sequence.
    `achieved` = ALGORITHMIC fp32 flops of the policy network (67 328 flop forward per row, 201 984
    flop forward+backward per sample) divided by the launch time; both kernels are latency-bound at this workload."""
    from xuance_amd import ops
    lr, mem, m = agent.learner, agent.memory, agent.model
    T, n, bs = agent.horizon_size, agent.n_envs, agent.batch_size
    fwd_flops_row = sum(2.0 * L.N * L.K for st in m.plan.stages for L in st)
    # (1) rollout kernel(s), without the GAE scan / counter bump of the rollout graph
    actor = agent._actor_rollout() is not None
    persistent = actor and getattr(agent, "persist_status", None) is not None
    kernel_only = lambda: agent._enqueue_rollout_fused(kernel_only=True)
    kernel_only()
    launches = 1 if persistent else (T if actor else T + 1)
    us_launch = _event_time_us(kernel_only, 5) / launches
    rows = n * T if persistent else (n if actor else 2 * n)   # rows through the whole network per launch (+ the sparse bootstrap rows)
    fl_launch = fwd_flops_row * rows
    name = ("xrl::actor_rollout_kernel + xrl::critic_values_kernel" if persistent else "xrl::actor_rollout_kernel") if actor \
        else "xrl::rollout_step_cartpole_kernel"
    r1 = {"bound": "mfma", "kernel": name, "achieved": round(fl_launch / us_launch / 1e6, 4),
          "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(fl_launch / us_launch / 1e6 / PEAK_FP32_MFMA_TFLOPS, 5),
          "traffic": _pmc_traffic("xrl::actor_rollout_kernel" if actor else name), "traffic_source": _PMC_SOURCE[0], "avg_launch_us": round(us_launch, 3),
          "algorithmic_flops_per_launch": fl_launch,
          "note": "latency-bound: %d rows x %.0f flop per launch (%s); see DESIGN.md section 3"
                  % (rows, fwd_flops_row, "%d vector steps of %d envs on the actor chain + their values as one batched pass" % (T, n)
                     if persistent else "one vector step")}
    # (2) fused minibatch kernel, timed INSIDE the real minibatch sequence (nb x [minibatch kernel, optimiser launch]) with
    #     an event pair around every minibatch launch.  Timed alone, back to back, the kernel re-reads parameters that are
    #     still in L2 and comes out ~10 % faster than what rocprofv3 sees in the loop.
    f = mem.soa.fields
    nb = agent.idx.shape[0]

    def mb(k):
        ops.ppo_fused_minibatch(m.plan, params=m.params.flat, params_t=lr.params_t, cache_image=lr.cache_image,
                                f_obs=f["observations"], f_act=f["actions"], f_ret=f["returns"], f_adv=f["advantages"],
                                f_logp=f["aux_old_logp"], idx=agent.idx[k], stats=lr.stats[k], slabs=lr.fslabs,
                                partials=lr.fpartials, diag=None, slab_stride=lr.slab_stride,
                                l0_fold_off=lr.fold[0] if lr.fold else 0, M=bs, n_envs=n, T=T, D=4,
                                frag_image=lr.frag, f_packed=lr.packed, f_rows=lr.rows[k * bs * 8:(k + 1) * bs * 8],
                                frag16=lr.frag16 if (getattr(lr, "pair", False) and getattr(lr, "frag16", None) is not None) else None,
                                pad0=(66 if getattr(lr, "chain", False) else 64) if getattr(lr, "pair", False) else 0,
                                A=m.action_dim, clip_range=lr.clip_range, vf_coef=lr.vf_coef, ent_coef=lr.ent_coef)
    r2 = None
    if lr.fused_eligible(mem) and getattr(lr, "rows", None) is not None and getattr(lr, "opt_sync", None) is not None:
        clip = lr.grad_clip_norm if lr.use_grad_clip else 0.0
        opt = lr.optimizer
        def ra():
            ops.reduce_adam(lr.fslabs, getattr(lr, "n_slabs", lr.n_tiles), lr.slab_stride, m.params.flat, opt.grad, opt.m, opt.v, m.params.P,
                            opt.state, lr.sumsq, clip, lr._mirrors, lr.opt_sync, fold=lr.fold)
        # the real minibatch sequence [minibatch kernel, reduce + Adam] x nb, bracketed twice with HIP events: once around
        # every PAIR, once around the optimiser launch of every pair only; the difference of the medians is the minibatch
        # kernel in its real position (cold slabs, parameters just rewritten), the cost of the event packets cancels.  Two
        # rollouts are enqueued first so that the host has queued the whole sequence before the device gets to it.
        # (Earlier versions: graphs of pairs minus a graph of optimiser launches run alone -- alone they read warm slabs,
        # the difference moved by +-4 us between runs; one bracket around the kernel minus an empty bracket -- an empty
        # bracket costs 5.2 us, more than the two event packets add around a kernel.)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nb)]

        def brackets(around_pair):
            out = []
            for _ in range(3):
                torch.cuda.synchronize()
                agent.rollout(); agent.rollout()
                for k in range(nb):
                    if around_pair:
                        evs[k][0].record(); mb(k); ra(); evs[k][1].record()
                    else:
                        mb(k); evs[k][0].record(); ra(); evs[k][1].record()
                torch.cuda.synchronize()
                out += [x.elapsed_time(y) * 1e3 for x, y in evs]
            out.sort()
            return out[len(out) // 2]
        us_pair, us_opt = brackets(True), brackets(False)
        us_mb = us_pair - us_opt
        fl_mb = 3.0 * fwd_flops_row * bs
        n_mb = nb                                          # agent.idx holds n_epochs x n_minibatch index rows
        bx = bool(lr.fold) and getattr(lr, "pair", False) and getattr(lr, "frag16", None) is not None
        # (role-split family, 64-row tiles at the headline size: the 128-wide products as exact 3-way bf16 splits, csrc/ppo_trunk_bx.hip)
        kname = ("xrl::ppo_trunk_bx_kernel" if bx else "xrl::ppo_trunk_kernel") if lr.fold else "xrl::ppo_fused_kernel"
        r2 = {"bound": "mfma", "kernel": kname, "achieved": round(fl_mb / us_mb / 1e6, 3),
              "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(fl_mb / us_mb / 1e6 / PEAK_FP32_MFMA_TFLOPS, 4),
              "traffic": _pmc_traffic(kname), "traffic_source": _PMC_SOURCE[0], "avg_launch_us": round(us_mb, 3),
              "algorithmic_flops_per_launch": fl_mb, "launches_per_step": n_mb, "us_per_step": round(us_mb * n_mb, 1),
              "note": "%d rows x %.0f flop (forward + backward) per launch; avg_launch_us = median HIP-event bracket around "
                      "[minibatch kernel, optimiser launch] inside the real sequence (%.1f us) minus the median bracket around "
                      "the optimiser launch in the same position (%.1f us); see DESIGN.md section 3"
                      % (bs, 3.0 * fwd_flops_row, us_pair, us_opt)}
        if bx:
            # what the matrix pipe executes: six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block of each of the three 128-wide
            # products (the exact 3-way split of both operands, the six largest of the nine part products), float32 accumulation
            big = 3.0 * 2.0 * 128 * 256 * bs
            r2["matrix_pipe"] = {"instruction": "v_mfma_f32_32x32x16_bf16 x 6 per fp32 32x32x16 block (exact 3-way bf16 split of both operands; "
                                                "dropped part products <= 2^-23 per scalar product), float32 accumulate",
                                 "executed_bf16_flops_per_launch": 6.0 * big, "executed_TFLOPs": round(6.0 * big / us_mb / 1e6, 2),
                                 "bf16_dense_peak_TFLOPs": 2500.0, "frac_of_bf16_peak": round(6.0 * big / us_mb / 1e6 / 2500.0, 4),
                                 "pipe_cycles_vs_fp32_instruction": "6 x 32 = 192 instead of 8 x 64 = 512 per block"}
            r2["note"] += ("; `peak` stays the float32 matrix peak the ALGORITHMIC float32 flops are priced against -- the products run as "
                           "bf16 split triples (matrix_pipe), results within float32 rounding of the fp32 instruction's")
    r1.update(launches_per_step=launches, us_per_step=round(us_launch * launches, 1))
    return r1, r2


def hbm_kernels(agent):
    """What the north-star asks rocprof to report beside the roofline of the dominant kernel: achieved HBM GB/s of the GAE / TD
    kernels and the matrix-pipe utilisation of the mixer GEMM.  gae_relay_kernel is timed LIVE (HIP events on the launch stream,
    algorithmic 20 B per transition); everything counter-based is COPIED from the latest committed rocprofv3 --pmc
    passes of the commands named (profiles/: separate passes, FETCH_SIZE corrected for gfx950 as MI355X_MICROARCH.md prescribes) --
    counters cannot be read from inside the process."""
    import glob
    from xuance_amd import ops
    out = {}
    try:
        f, T, n = agent.memory.soa.fields, agent.horizon_size, agent.n_envs
        gae = lambda: ops.gae_scan(f["rewards"], f["values"], f["terminals"], f["bootv"], f["seg"], f["advantages"], f["returns"],
                                   agent.gamma, agent.gae_lam, agent.memory.use_gae)
        gae()
        us = _event_time_us(gae, 50)
        out["gae_relay_kernel"] = {"bound": "hbm", "algorithmic_bytes_per_launch": 20 * n * T + 4 * n, "avg_launch_us": round(us, 3),
                                   "achieved_GBps": round((20 * n * T + 4 * n) / us / 1e3, 1), "peak_GBps": 8000.0, "measured": "live (HIP events)",
                                   "note": "%d envs x %d steps: 1.3 MB per launch is far below one launch's worth of HBM work (launch-latency-bound)" % (n, T)}
    except Exception as ex:                                    # noqa: BLE001
        out["gae_relay_kernel"] = {"error": repr(ex)[:200]}
    try:
        sw = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scale_sweep*.jsonl")))[-1]
        rows = [json.loads(l) for l in open(sw) if l.startswith("{")]
        g = max((r for r in rows if r.get("kernel") == "gae_scan"), key=lambda r: r["algorithmic_MB"])
        out["gae_relay_kernel_at_scale"] = {"n_envs": g["n_envs"], "T": g["T"], "avg_launch_us": g["us"], "achieved_GBps": g["GBps"],
                                            "peak_GBps": 8000.0, "frac": round(g["GBps"] / 8000.0, 3),
                                            "source": "profiles/" + os.path.basename(sw) + " (committed, tools/scale_sweep.py; not measured in this run)"}
    except Exception as ex:                                    # noqa: BLE001
        out["gae_relay_kernel_at_scale"] = {"error": repr(ex)[:200]}
    for key, pat, kernel in (("dqn_tail_td_kernel", "r*_dqn_c3_pmc.json", "xrl::dqn_tail_td_kernel"),
                             ("qmix_prefetch_kernel", "r*_qmix_gru_pmc.json", "xrl::qmix_prefetch_kernel"),
                             ("mixer_gemm", "r*_qmix_gru_pmc.json", "xrl::gemm_f32_kernel<0, true, true>")):
        try:
            path = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1]
            k = json.load(open(path))["kernels"][kernel]
            e = {"kernel": kernel, "avg_launch_us": k["avg_us"], "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                 "achieved_GBps": k["hbm_GBps"], "peak_GBps": 8000.0,
                 "source": "profiles/" + os.path.basename(path) + " (committed rocprofv3 --pmc passes; not measured in this run)"}
            if key == "mixer_gemm":
                e["mfma_busy_frac"] = k["mfma_busy_frac"]
                e["note"] = "the recurrent QMIX update's 1 920-row mixer / 5 856-row agent products (gemm_f32_kernel, fp32 MFMA)"
            out[key] = e
        except Exception as ex:                                # noqa: BLE001
            out[key] = {"error": repr(ex)[:200]}
    return out


def cpu_baseline(n_envs, horizon, budget_s=20.0):
    """The oracle's CPU port of the same loop (oracle/cpu_agent.py), timed on this host."""
    from oracle import cpu_agent
    try:
        from threadpoolctl import threadpool_info
        threads = max([d.get("num_threads", 1) for d in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    r = cpu_agent.run_ppo_cartpole(n_envs=n_envs, horizon=horizon, n_rollouts=6, time_budget_s=budget_s)
    return {"value": round(r["env_steps"] / r["seconds"], 1), "unit": "env-steps/s", "cores": int(threads),
            "kind": "port",
            "sample": "%d rollout(s) of %d envs x %d steps incl. 64 minibatch updates each, oracle/cpu_agent.py "
                      "(NumPy, %d BLAS threads, %.1f s)" % (r["rollouts"], n_envs, horizon, threads, r["seconds"])}


def box_yardstick():
    """Two fixed measurements of the BOX this run landed on -- not of this library: a 256 MB device copy (HBM rate) and a
    chain of 400 one-thread launches on one stream (what a small dependent launch costs here).  Fresh boxes of the pool
    differ (the same library gave a 147-149 us DQN-C3 update graph on some and 178-179 us on
    others); every launch-bound line of this JSON scales with the second number, so it is recorded next to them."""
    from xuance_amd import ops
    x = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    c = torch.zeros(1, dtype=torch.int32, device="cuda")
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(3):
        y.copy_(x)
    for _ in range(50):
        ops.counter_add(c, 1)
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(10):
        y.copy_(x)
    e[1].record()
    e[2].record()
    for _ in range(400):
        ops.counter_add(c, 1)
    e[3].record()
    torch.cuda.synchronize()
    out = {"copy_GBps": round(10 * 2 * x.numel() * 4 / (e[0].elapsed_time(e[1]) * 1e-3) / 1e9, 1),
           "small_launch_us": round(e[2].elapsed_time(e[3]) * 1e3 / 400, 3)}
    # (round 6, review item 8) the shader clock this box sustains under matrix-core load: shader cycles / 100 MHz wall ticks of a chain of
    # dependent fp32 MFMAs on every CU (tools/csrc/probe.hip: mfma_chain_kernel) -- every compute-bound line scales with it
    try:
        import ctypes as C
        from tools import probe_lib
        o = torch.zeros(2, dtype=torch.int64, device="cuda"); sink = torch.zeros(64 * 1024, device="cuda")
        for _ in range(2):
            probe_lib.call("xrl_probe_mfma_chain", 20000, 1024, C.c_void_p(o.data_ptr()), C.c_void_p(sink.data_ptr()),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
        cyc, ticks = o.tolist()
        out["shader_clock_GHz_mfma_load"] = round(cyc / (ticks / 100e6) / 1e9, 3)
    except Exception as ex:                                        # noqa: BLE001  (diagnostics library not built: no yardstick)
        out["shader_clock_GHz_mfma_load"] = None
        out["shader_clock_note"] = repr(ex)[:120]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-envs", type=int, default=256, help="envs PER GPU")
    ap.add_argument("--horizon", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC pass instead of two rocprofv3 passes of a 3-step child run")
    ap.add_argument("--no-secondary", action="store_true", help="skip the PPO-16-envs / QMIX-3m / eager-PyTorch lines")
    ap.add_argument("--no-role-split", action="store_true", help="the any-shape minibatch kernel (one workgroup per tile, ppo_fused_kernel) instead of the role-split family")
    ap.add_argument("--workload", choices=bw.WORKLOADS, default="c2", help="which BASELINE configuration is the main line")
    ap.add_argument("--grad-path", choices=("measure", "auto", "exchange", "captured", "cut"), default="measure",
                    help="N > 1: how the ranks average gradients; measure (default) = time every usable way, adopt the fastest")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: this process becomes the launcher of its N ranks (one process per GPU, rendezvous on
        # 127.0.0.1 -- the reference's launch contract, xuance/torch/utils/operations.py:11-28: RANK / LOCAL_RANK / WORLD_SIZE from
        # the environment) and returns their exit code; the driver's own `python -m torch.distributed.run ... bench.py --gpus N`
        # arrives here with WORLD_SIZE set and runs as a rank.
        import socket
        import subprocess
        if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus and not os.environ.get("XRL_BENCH_SHARE_GPUS"):
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (XRL_BENCH_SHARE_GPUS=1 lets ranks share devices: a "
                             "functional check, not a measurement)" % (args.gpus, torch.cuda.device_count()))
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
            env.setdefault("XRL_DIST_BACKEND", "gloo")      # (ranks sharing a device: RCCL refuses duplicate GPUs; the functional check uses gloo)
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks -- refusing to report a line whose n_gpus "
                         "is not what was asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    local_rank %= torch.cuda.device_count()            # (ranks share a device only in the 2-rank gloo test on a 1-GPU box)
    os.environ["LOCAL_RANK"] = str(local_rank)
    torch.cuda.set_device(local_rank)
    if world > 1:
        from xuance_amd.dist import init_distributed_mode
        init_distributed_mode(os.environ.get("XRL_DIST_BACKEND", "nccl"))     # "nccl" = RCCL over xGMI
    import torch.distributed as dist
    device = torch.device("cuda", local_rank)
    c2 = args.workload == "c2"
    n_envs = args.n_envs if c2 else None

    path, paths_ms = "auto", None
    if world > 1:
        if args.grad_path == "measure":
            path, paths_ms = bw.measure_paths(args.workload, world, rank, device, n_envs, args.horizon)
        else:
            path = args.grad_path
    extra = {"use_role_split_update": not args.no_role_split} if c2 else None
    runner = bw.Runner(args.workload, world, rank, path, n_envs, args.horizon, extra=extra)
    agent = runner.agent
    elapsed, env_steps = bw.timed(runner, args.steps, args.warmup, world)
    info = runner.info or {}

    out = {"metric": runner.metric, "value": round(env_steps / elapsed, 1),
           "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": runner.describe, "env": runner.env_name, "parallelism": "dp%d" % world,
                      "env_steps_per_step": env_steps // args.steps, "rollout_mode": runner.rollout_mode(),
                      "last_info": {k: round(float(v), 6) for k, v in info.items() if isinstance(v, (int, float))}}}
    lr_ = getattr(agent, "learner", None)
    if c2 and getattr(lr_, "frag16", None) is not None:
        # operands, accumulators and results of the timed region are float32; what the key does NOT say: how the minibatch launch forms
        # its three 128-wide float32 products on the matrix cores (no value is rounded to bf16 anywhere)
        out["dtype_note"] = ("f32 end to end.  The update launch (xrl::ppo_trunk_bx_kernel) forms its 128-wide float32 products as exact 3-way bf16 "
                             "splits of both operands (x == h + m + l bit for bit), six v_mfma_f32_32x32x16_bf16 per 32x32x16 block with float32 "
                             "accumulation; the three dropped part products are <= 2^-23 of a scalar product.  Same reference fixtures at the same "
                             "1e-5 as the float32-instruction kernel (config use_split_products: False selects that one); DESIGN.md section 3 'Round 6' (j)")
    if world > 1:      # which way averaged the gradients in the timed region, what every usable way cost, and a sign of life of RCCL
        out["config"]["gradient_average"] = runner.gradient_average()
        out["config"]["gradient_path"] = path
        out["config"]["gradient_paths_ms"] = paths_ms
        out["config"]["rccl_world"] = bw.rccl_world(world)
        out["config"]["backend"] = dist.get_backend()
        # (round 6: also at the top level of the line, so that one SCALE run of the driver records every path's cost per N)
        out["gradient_paths_ms"], out["rccl_world"], out["gradient_path"] = paths_ms, out["config"]["rccl_world"], path
        if out["config"]["rccl_world"] != world:
            raise SystemExit("bench.py: the RCCL all-reduce over the %d ranks counted %s participants" % (world, out["config"]["rccl_world"]))
    if rank == 0:
        try:
            out["config"]["box"] = box_yardstick()
        except Exception as ex:                                    # (a yardstick must never cost the line)
            out["config"]["box"] = {"error": repr(ex)[:200]}
    # the update-phase rate separately (transitions consumed per second by GAE + sampling + the
    # minibatch updates), so that the simulator's share is separable.  Timed after the contract region, same graphs.
    phases = None
    if world == 1 and c2 and not args.no_roofline:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            agent.rollout()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(5):
            agent.update()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        n_tr = args.n_envs * args.horizon
        phases = {"rollout_ms": round((t2 - t1) / 5 * 1e3, 4), "update_ms": round((t3 - t2) / 5 * 1e3, 4),
                  "rollout_env_steps_per_s": round(n_tr * 5 / (t2 - t1), 1),
                  "update_transitions_per_s": round(n_tr * 5 / (t3 - t2), 1),
                  "update_sample_passes_per_s": round(n_tr * 8 * 5 / (t3 - t2), 1),
                  "note": "rollout = %d vector steps incl. GAE; update = 8 epochs x 8 minibatches over the same %d transitions"
                          % (args.horizon, n_tr)}
    nrank_secondary = {}
    if world > 1 and c2 and not args.no_secondary:
        # the two other node-sized BASELINE configurations on the same ranks (every rank takes part: collectives inside)
        runner.close()
        for w, st, wu in (("c4", 3, 1), ("qmix3m", 12, 3)):
            try:
                pw, pms = bw.measure_paths(w, world, rank, device, None, 256, passes=1 if w == "c4" else 4)
                r = bw.Runner(w, world, rank, pw)
                el, ns = bw.timed(r, st, wu, world)
                nrank_secondary[w] = {"workload": r.describe, "value": round(ns / el, 1), "unit": "env-steps/s", "n_gpus": world,
                                      "steps": st, "warmup": wu, "ms_per_step": round(el / st * 1e3, 4), "scaling": "weak",
                                      "env_steps_per_step": ns // st, "gradient_average": r.gradient_average(),
                                      "gradient_path": pw, "gradient_paths_ms": pms, "rollout_mode": r.rollout_mode()}
                r.close()
            except Exception as ex:                          # noqa: BLE001
                nrank_secondary[w] = {"error": repr(ex)[:300]}
    if rank == 0:
        if nrank_secondary:
            out["secondary"] = nrank_secondary
        if phases is not None:
            out["phases"] = phases
        pmc_note = None
        if world == 1 and c2 and not args.no_roofline and not args.no_live_pmc:
            try:
                pmc_note = live_pmc()
            except Exception as ex:                          # noqa: BLE001  (a profiler problem must never cost the line)
                pmc_note = repr(ex)[:200]
        if c2 and not args.no_roofline:
            # `roofline` = the kernel with the largest share of the step (what rocprofv3 --stats puts first: the fused
            # minibatch kernel at the headline workload), the other of the two rides along under its own key
            first, second = kernel_rooflines(agent)
            if second is not None and second["us_per_step"] > first["us_per_step"]:
                out["roofline"], out["roofline_rollout_kernel"] = second, first
            else:
                out["roofline"] = first
                if second is not None:
                    out["roofline_update_kernel"] = second
            out["hbm_kernels"] = hbm_kernels(agent)
            if pmc_note:
                out["roofline"]["live_pmc_note"] = pmc_note
        if world == 1 and c2 and not args.no_cpu_baseline:
            # cpu_baseline: the reference's own CPU torch path (kind "reference", timed where the reference exists);
            # cpu_port: the oracle's NumPy port of the same loop, timed live on THIS host's cores
            out["cpu_baseline"] = reference_cpu_baseline("ppo_cartpole", str(args.n_envs))
            out["cpu_port"] = cpu_baseline(args.n_envs, args.horizon)
            # (round 6: the one SAME-HOST CPU figure also rides inside cpu_baseline, next to the reference's number from the build host)
            if isinstance(out["cpu_baseline"], dict):
                out["cpu_baseline"]["same_host_port"] = out["cpu_port"]
        if world == 1 and c2 and not args.no_secondary:
            sec = {}
            try:
                import eager_torch_ppo
                out["gpu_eager_baseline"] = eager_torch_ppo.measure(args.n_envs, args.horizon, steps=2, warmup=1)
            except Exception as ex:                          # noqa: BLE001
                out["gpu_eager_baseline"] = {"error": repr(ex)}
            try:
                import bench_secondary as bs
                sec["ppo_cartpole_16_envs"] = bs.ppo_small(make_config, kernel_rooflines, 16, args.horizon,
                                                           ref=reference_cpu_baseline("ppo_cartpole", "16"))
                sec["ppo_acrobot_256_envs"] = bs.ppo_acrobot(make_config)
                # (the reference's default size for this yaml: configs/ppo/classic_control/Acrobot-v1.yaml, parallels 10)
                sec["ppo_acrobot_10_envs"] = bs.ppo_acrobot(make_config, n_envs=10, steps=10, warmup=3)
                sec["qmix_3m_ff"] = bs.qmix_3m(False, ref=reference_cpu_baseline("qmix_3m_ff"))
                sec["qmix_3m_gru"] = bs.qmix_3m(True, ref=reference_cpu_baseline("qmix_3m_gru"))
                # the two remaining BASELINE configs at their per-GPU shapes (the reference's CPU time exists per update only:
                # profiles/ref_cpu_baseline.json; no simulator for either is installed, the providers are synthetic)
                sec["ppo_halfcheetah_shape_c4"] = bs.ppo_c4(ref=reference_cpu_baseline("ppo_halfcheetah_shape_c4"))
                sec["dqn_atari_shape_c3"] = bs.dqn_c3(ref=reference_cpu_baseline("dqn_atari_shape_c3"))
                sec["ppo_atari_shape"] = bs.ppo_atari(ref=reference_cpu_baseline("ppo_atari_shape"))          # (configs/ppo/atari.yaml: the uint8 rollout buffer's user)
            except Exception as ex:                          # noqa: BLE001
                sec["error"] = repr(ex)
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
