"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares, ctypes
structs match the C layout, the parameter layout / state_dict order of the nets, buffer path bookkeeping, the
plain-C GAE restatement, and the multi-rank plumbing over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, sub, free_port


def test_library_exports_every_declared_symbol():
    from xuance_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "xrl_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(xrl_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = _lib.load()                         # dlopen works without a GPU; no compute call is made here
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/xrl_hip.h but not exported"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert b"gfx950" in lib.xrl_version()


def test_ctypes_structs_match_c_layout():
    from xuance_amd import _lib
    pairs = {"xrl_field_t": _lib.Field, "xrl_gemm_t": _lib.Gemm, "xrl_ppo_loss_t": _lib.PpoLoss,
             "xrl_adam_state_t": _lib.AdamState, "xrl_rms_t": _lib.Rms, "xrl_sample_t": _lib.Sample,
             "xrl_cartpole_t": _lib.CartPole, "xrl_poststep_t": _lib.PostStep, "xrl_egreedy_t": _lib.EGreedy,
             "xrl_mirrors_t": _lib.Mirrors, "xrl_exchange_t": _lib.Exchange, "xrl_marl_gate_t": _lib.MarlGate,
             "xrl_ppo_wide_t": _lib.PpoWide, "xrl_wide_act_t": _lib.WideAct, "xrl_conv_t": _lib.Conv, "xrl_classic_t": _lib.Classic,
             "xrl_dqn_head_td_t": _lib.DqnHeadTd, "xrl_dqn_tail_td_t": _lib.DqnTailTd, "xrl_dqn_act_tail_t": _lib.DqnActTail,
             "xrl_marl_act_t": _lib.MarlAct, "xrl_marl_act_gru_t": _lib.MarlActGru, "xrl_rollout_run_t": _lib.RolloutRun,
             "xrl_rollout_wide_t": _lib.RolloutWide,
             # round 6
             "xrl_act_tail_t": _lib.ActTail, "xrl_mlp_chain_job_t": _lib.MlpChainJob, "xrl_mlp_chain_t": _lib.MlpChain,
             "xrl_ppo_act_tail_t": _lib.PpoActTail, "xrl_opt_chain_t": _lib.OptChain, "xrl_qmix_phase_t": _lib.QmixPhase,
             "xrl_qa_image_t": _lib.QaImage, "xrl_ppo_fused_t": _lib.PpoFused}
    for extra in ("xrl_dqn_td_t", "xrl_qmix_t"):
        cls = getattr(_lib, {"xrl_dqn_td_t": "DqnTd", "xrl_qmix_t": "Qmix"}[extra], None)
        if cls is not None:
            pairs[extra] = cls
    src = '#include "xrl_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){\n'
    for cname in pairs:
        src += f'printf("{cname} %zu\\n", sizeof({cname}));\n'
    src += 'printf("adam.base_lr %zu\\n", offsetof(xrl_adam_state_t, base_lr));\n'
    src += 'printf("act.post %zu\\n", offsetof(xrl_wide_act_t, post));\nprintf("act.xchg %zu\\n", offsetof(xrl_wide_act_t, xchg));\n'
    src += 'printf("wide.obs %zu\\n", offsetof(xrl_ppo_wide_t, obs));\nprintf("wide.dbg %zu\\n", offsetof(xrl_ppo_wide_t, dbg));\n'
    src += 'printf("loss.M %zu\\n", offsetof(xrl_ppo_loss_t, M));\n'
    offs = {"xrl_mirrors_t": (_lib.Mirrors, ("tick", "part_out", "tick_inc", "part_rows", "alt_lo", "alt_hi", "alt_split")),   # fields added in rounds 3, 4
            "xrl_poststep_t": (_lib.PostStep, ("ret_count", "n", "gamma", "pg_bootv")),
            "xrl_marl_gate_t": (_lib.MarlGate, ("ring", "reset_rule", "done", "ptr_size", "buffer_size", "end_step")),   # round 5
            "xrl_egreedy_t": (_lib.EGreedy, ("eps", "seed", "step_dev")), "xrl_marl_act_t": (_lib.MarlAct, ("eps", "seed")),
            "xrl_marl_act_gru_t": (_lib.MarlActGru, ("eps_dev", "step", "eps", "lds_staged")),
            "xrl_act_tail_t": (_lib.ActTail, ("heads", "ldh", "boot_rows", "env_kind", "act_actor", "sample", "classic", "cartpole")),
            "xrl_mlp_chain_job_t": (_lib.MlpChainJob, ("layers", "n_layers", "level_width", "out", "ld_out", "ldx", "M")),
            "xrl_mlp_chain_t": (_lib.MlpChain, ("n_jobs", "tile0")),
            "xrl_qa_image_t": (_lib.QaImage, ("image_floats", "lds_bytes", "interleaved")),
            "xrl_dqn_tail_td_t": (_lib.DqnTailTd, ("partials", "M", "act", "gamma", "huber_delta", "slabs", "slab_stride", "off_b2")),
            "xrl_dqn_head_td_t": (_lib.DqnHeadTd, ("partials", "act", "gamma", "huber_delta")),                 # round 5: Huber switch
            "xrl_dqn_td_t": (_lib.DqnTd, ("partials", "gamma", "dueling", "huber_delta")),
            "xrl_dqn_act_tail_t": (_lib.DqnActTail, ("step_dev", "seed", "step", "n", "act", "eps", "eps_sched", "eps_kstar", "eps_start", "eps_delta")),
            "xrl_ppo_wide_t": (_lib.PpoWide, ("rows_g2", "rows_h1", "rows_ld")),
            "xrl_rollout_run_t": (_lib.RolloutRun, ("act", "flags", "gamma", "seed", "step", "step_dev", "obs_raw", "cp_stats", "f_val", "xchg", "dbg")),
            "xrl_rollout_wide_t": (_lib.RolloutWide, ("log_std_off", "H", "flags", "gamma", "seed", "env_step", "env_step_dev", "obs_raw", "ret_var",
                                                      "env_stats", "Bmat", "f_seg", "raw_rew", "xchg", "dbg"))}
    for cname, (cls, fields) in offs.items():
        for f in fields:
            src += f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));\n'
    src += 'return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "sz.c"), os.path.join(d, "sz")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = dict(line.split() for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
    for cname, (cls, fields) in offs.items():
        for f in fields:
            assert int(out[f"{cname}.{f}"]) == getattr(cls, f).offset, (cname, f)
    assert int(out["adam.base_lr"]) == _lib.AdamState.base_lr.offset
    assert int(out["loss.M"]) == _lib.PpoLoss.M.offset
    assert int(out["wide.obs"]) == _lib.PpoWide.obs.offset and int(out["wide.dbg"]) == _lib.PpoWide.dbg.offset
    assert int(out["act.post"]) == _lib.WideAct.post.offset and int(out["act.xchg"]) == _lib.WideAct.xchg.offset


@pytest.mark.parametrize("dist,args", [("categorical", (4, 2, "categorical", (128,), (128,), (128,), "leaky_relu")),
                                       ("gaussian", (17, 6, "gaussian", (), (64, 64), (64, 64), "relu", "tanh"))])
def test_net_layout_and_state_dict_order(dist, args):
    from xuance_amd.nets import ActorCriticNet
    g = load_golden(f"ppo_{dist}")
    net = ActorCriticNet(*args, device="cpu")
    assert list(net.ref_order) == [str(n) for n in g["param_names"]]       # the reference's state_dict order
    init = sub(g, "init")
    net.load_state_dict(init)
    sd = net.state_dict()
    for k, v in init.items():
        assert sd[k].shape == v.shape and np.array_equal(sd[k].numpy(), v)
    P = net.params
    for name in P.names:
        assert P.offsets[name] % 4 == 0                                     # 16-byte aligned tensors
    a0 = "actor.logits.0" if dist == "categorical" else "actor.mu.0"
    wa, wc = P.offsets[a0 + ".weight"], P.offsets["critic.values.0.weight"]
    n_a = int(np.prod(P.shapes[a0 + ".weight"]))
    assert wc == wa + n_a                                                   # stacked [Wa;Wc] is contiguous
    assert P.offsets["critic.values.0.bias"] == P.offsets[a0 + ".bias"] + P.shapes[a0 + ".bias"][0]
    n_params = sum(int(np.prod(v.shape)) for v in init.values())
    assert n_params == (34051 if dist == "categorical" else n_params) and P.P >= n_params


def test_orthogonal_init_follows_reference_order():
    from xuance_amd.nets import ActorCriticNet
    torch.manual_seed(7)
    net = ActorCriticNet(4, 2, "categorical", (128,), (128,), (128,), "leaky_relu", device="cpu")
    torch.manual_seed(7)
    ws = [torch.nn.init.orthogonal_(torch.empty(s)) for s in [(128, 4), (128, 128), (2, 128), (128, 128), (1, 128)]]
    sd = net.state_dict()
    for k, w in zip([k for k in net.ref_order if k.endswith("weight")], ws):
        assert torch.equal(sd[k], w), k
    assert all(float(sd[k].abs().sum()) == 0 for k in net.ref_order if k.endswith("bias"))


def test_buffer_path_bookkeeping_on_host():
    """finish_path only records (bootv, seg); check the records against the call pattern incl. the corner cases."""
    from xuance_amd.memory import HipOnPolicyBuffer
    from xuance_amd.spaces import Box, Discrete
    buf = HipOnPolicyBuffer(Box(-1, 1, (3,)), Discrete(2), {"old_logp": ()}, 4, 6, device="cpu")
    buf.ptr, buf.size = 3, 3
    buf.finish_path(0.0, 0)                       # Python float -> float64-carry flag
    buf.finish_path(np.float32(1.5), 1)           # float32 bootstrap
    buf.finish_path(torch.tensor(2.5), 2)
    assert buf._seg_h[2].tolist() == [3, 1, 1, 0] and buf._bootv_h[2].tolist() == [0.0, 1.5, 2.5, 0.0]
    assert buf.start_ids.tolist() == [3, 3, 3, 0]
    buf.finish_path(9.0, 0)                       # empty slice [3,3): no-op
    assert buf._seg_h.sum() == 5
    buf.ptr, buf.size = 0, 6                      # buffer full, ptr wrapped
    for i in range(4):
        buf.finish_path(np.float32(0.25), i)
    assert buf._seg_h[5].tolist() == [1, 1, 1, 1] and buf.start_ids.tolist() == [0, 0, 0, 0]
    buf.finish_path(0.0, 3)                       # second call over the whole row: ONE path [0,6) (reference quirk)
    assert buf._seg_h[:, 3].tolist() == [0, 0, 0, 0, 0, 3]
    buf.finish_path(0.0, 0)
    assert buf._seg_h[:, 0].tolist() == [0, 0, 0, 0, 0, 3]     # earlier boundary at t=2 dropped, like the reference
    with pytest.raises(AssertionError):
        HipOnPolicyBuffer(Box(-1, 1, (3,)), Discrete(2), None, 4, 6, device="cpu").sample(np.arange(4))


def test_c_restatement_matches_numpy_oracle(oracle):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libgae_ref.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.gae_finish_path_ref.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double, fp, fp]
    rng = np.random.default_rng(0)
    for L in (1, 2, 17, 256):
        for pyfloat in (0, 1):
            r, v = rng.standard_normal(L).astype(np.float32), rng.standard_normal(L).astype(np.float32)
            d = (rng.random(L) < 0.1).astype(np.float32)
            val = 0.0 if pyfloat else np.float32(rng.standard_normal())
            ret, adv = np.zeros(L, np.float32), np.zeros(L, np.float32)
            lib.gae_finish_path_ref(r.ctypes.data_as(fp), v.ctypes.data_as(fp), d.ctypes.data_as(fp), L, float(val),
                                    pyfloat, 0.98, 0.95, ret.ctypes.data_as(fp), adv.ctypes.data_as(fp))
            eret, eadv = oracle.gae_finish_path(r, v, d, val, 0.98, 0.95)
            assert np.array_equal(adv, eadv) and np.array_equal(ret, eret)


def _gloo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from xuance_amd import dist as xd
    import torch.distributed as dist
    r, w, _ = xd.init_distributed_mode("gloo")
    g = torch.full((1000,), float(r + 1))
    xd.allreduce_mean_(g)                                        # DDP-style gradient mean as one flat message
    p = torch.arange(10, dtype=torch.float32) * (r + 1)
    xd.broadcast_(p, 0)                                           # rank 0's parameters everywhere
    lo, hi = xd.shard_range(512)
    t = torch.tensor([0.5 + r], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # bench.py's max-over-ranks timing
    # tensor_statistics.py:48-58: means and variances averaged over the ranks, counts summed -- one packed message here
    bm, bv, bc = torch.tensor([1.0, 2.0]) * (r + 1), torch.tensor([0.5, 0.25]) * (r + 1), torch.tensor(8.0 * (r + 1))
    xd.allreduce_moments_(bm, bv, bc)
    assert bm.tolist() == [1.5, 3.0] and bv.tolist() == [0.75, 0.375] and float(bc) == 24.0
    q.put((r, w, float(g[0]), p.tolist(), (lo, hi), float(t)))
    xd.barrier()
    dist.destroy_process_group()


def test_multi_rank_plumbing_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, w, gmean, params, shard, tmax in res:
        assert w == 2 and gmean == 1.5 and params == list(map(float, range(10))) and tmax == 1.5
    assert res[0][4] == (0, 256) and res[1][4] == (256, 512)


def _gloo_ddp_worker(rank, world, port, q, kind):
    """One rank of the two-process replay of tests/golden/ddp2_<kind>.npz on CPU: the oracle's gradient of THIS rank's batch, the ranks'
    mean through the product's own collective helper (dist.allreduce_mean_ on ONE flat float32 message: what every learner calls
    between its slab reduction and its optimiser step), then clip + Adam as the oracle restates them."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import traceback
    import numpy as np
    import torch
    import torch.distributed as dist
    from conftest import load_golden, sub, LearnerFixtureCheck
    from oracle import xrl_oracle as oracle
    from xuance_amd import dist as xd
    torch.set_num_threads(2)
    xd.init_distributed_mode("gloo")
    err = None
    try:
        g, d = load_golden(kind), load_golden("ddp2_" + kind)
        if kind == "dqn_mlp":
            lr, gamma, sync, gclip, use_clip, total = g["cfg"]
            clip = gclip if use_clip else None
            fb = lambda sd, b: oracle.dqn_forward_backward(sd, b, dict(gamma=gamma, double_q=False, huber_delta=0.0))
            copy_target = oracle.dqn_copy_target
        else:
            lr, gamma, sync, gclip, dq, total = g["cfg"]
            clip = gclip
            fb = lambda sd, b: oracle.qmix_forward_backward(sd, b, dict(gamma=gamma, double_q=bool(dq), use_actions_mask=True), group=str(g["group"]))
            copy_target = oracle.qmix_copy_target
        sd = {k: v.copy() for k, v in sub(g, "init").items()}
        names = [str(n) for n in g["param_names"]]
        opt = oracle.AdamOracle({k: sd[k] for k in names}, lr=lr, total_iters=int(total))
        merged = dict(d)
        merged.update({k: v for k, v in g.items() if k.startswith("init/")})
        chk = LearnerFixtureCheck(merged, sd, lr, total_iters=int(total))
        for u in range(int(d["n_updates"])):
            _, grads = fb(sd, sub(g, f"u{(rank + u) % 2}/batch"))
            order = [n for n in names if n in grads]
            flat = torch.from_numpy(np.concatenate([np.asarray(grads[n], np.float32).ravel() for n in order]))
            xd.allreduce_mean_(flat)                                 # the collective under test
            off, avg = 0, {}
            for n in order:
                k = grads[n].size
                avg[n] = flat[off:off + k].numpy().reshape(grads[n].shape).copy()
                off += k
            if clip is not None:
                oracle.AdamOracle.clip_grad_norm_(avg, clip)
            opt.step(avg)
            if (u + 1) % int(sync) == 0:
                copy_target(sd)
            chk.update(u, avg, sd)
        assert chk.replay_checked > 0
    except Exception:                                             # noqa: BLE001
        err = traceback.format_exc()
    q.put((rank, err, None if err else np.concatenate([sd[n].ravel() for n in names])))
    xd.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["dqn_mlp", "qmix_ff_double"])
def test_two_rank_gradient_mean_gloo_vs_reference_ddp(kind):
    """World-size-2 gloo run on CPU of the N-rank update rule against the REFERENCE's two-process DistributedDataParallel run
    (tests/golden/ddp2_*.npz, oracle/make_golden_ddp.py): each rank's own batch, the mean through xuance_amd.dist.allreduce_mean_, the
    averaged clipped gradients and the parameter steps of both updates within the fixture tolerances; replicas bit-identical."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_gloo_ddp_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, _ in res:
        assert err is None, f"rank {rank}:\n{err}"
    assert np.array_equal(res[0][2], res[1][2])


def test_checkpoint_layout_is_the_references(tmp_path):
    """(1) a `.pth` written by the REFERENCE's Learner.save_model (tests/golden/ppo_ckpt_ref.pth, produced by
    oracle/make_golden.py golden_checkpoint) loads into this engine's learner -- parameters, Adam moments, step, lr;
    (2) a file written by this engine has the reference's layout: torch.optim.Adam over parameters of the same shapes
    accepts its optimiser state and steps with it.  Host-side only (no kernels are launched)."""
    import os
    from argparse import Namespace
    import numpy as np
    import torch
    from conftest import load_golden, sub, GOLDEN
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import PPO_Learner
    g = load_golden("ppo_ckpt")
    cfg = Namespace(horizon_size=256, n_epochs=8, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.98,
                    learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5,
                    distributed_training=False, device="cpu", model_dir=str(tmp_path))
    net = ActorCriticNet(4, 2, "categorical", (128,), (128,), (128,), "leaky_relu", device="cpu")
    learner = PPO_Learner(cfg, net)
    ref_path = os.path.join(GOLDEN, "ppo_ckpt_ref.pth")
    learner.load_model(ref_path)
    ref = torch.load(ref_path, weights_only=True)
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), sub(g, "saved")[k]), k
    st = learner.optimizer.read()
    assert st.step == 2 and abs(st.last_lr - 4e-4) < 1e-12
    osd = learner.optimizer.state_dict()
    for i in range(len(net.ref_order)):
        assert torch.equal(osd["state"][i]["exp_avg"], ref["optimizer"]["state"][i]["exp_avg"])
        assert torch.equal(osd["state"][i]["exp_avg_sq"], ref["optimizer"]["state"][i]["exp_avg_sq"])
    # (2) our file -> the stock optimiser
    mine = str(tmp_path / "mine.pth")
    learner.save_model(mine)
    ck = torch.load(mine, weights_only=True)
    assert list(ck.keys()) == list(ref.keys()) and list(ck["policy"].keys()) == list(ref["policy"].keys())
    assert set(ck["optimizer"]["param_groups"][0]) == set(ref["optimizer"]["param_groups"][0])
    params = [torch.nn.Parameter(v.clone()) for v in ck["policy"].values()]
    opt = torch.optim.Adam(params, 1e-3, eps=1e-5)
    opt.load_state_dict(ck["optimizer"])
    for p in params:
        p.grad = torch.zeros_like(p)
    opt.step()
    assert float(opt.state[params[0]]["step"]) == 3.0 and opt.param_groups[0]["lr"] == 4e-4


def test_shm_subproc_vec_env_matches_in_process_stepping():
    """ShmSubprocVecEnv: 6 host CartPole envs in 2 worker processes, step data through the shared block --
    observation / reward / flags / reset_obs / episode statistics identical to stepping the same seeded envs in this
    process with the reference worker's auto-reset rule (subproc_vec_env.py:9-14), through many episode ends."""
    import numpy as np
    from xuance_amd.envs import ShmSubprocVecEnv, NumpyCartPoleEnv
    n, steps = 6, 400
    venv = ShmSubprocVecEnv([NumpyCartPoleEnv] * n, env_seed=11, in_series=3, device="cpu")
    try:
        assert venv.num_envs == n and venv.max_episode_steps == 500 and venv.n_remotes == 2
        local = [NumpyCartPoleEnv(env_seed=11 + i) for i in range(n)]
        obs, infos = venv.reset()
        ref = np.stack([e.reset()[0] for e in local])
        assert np.array_equal(obs, ref) and len(infos) == n
        rng = np.random.default_rng(0)
        ends = 0
        for t in range(steps):
            acts = rng.integers(0, 2, n)
            o, r, term, trunc, infos = venv.step(acts)
            for i, e in enumerate(local):
                eo, er, et, etr, einfo = e.step(acts[i])
                assert np.array_equal(o[i], eo) and r[i] == er and term[i] == et and trunc[i] == etr
                assert infos[i]["episode_step"] == einfo["episode_step"] and infos[i]["episode_score"] == einfo["episode_score"]
                if et or etr:
                    ends += 1
                    assert np.array_equal(infos[i]["reset_obs"], e.reset()[0])
                else:
                    assert "reset_obs" not in infos[i]
        assert ends > 10
    finally:
        venv.close()
    assert all(not p.is_alive() for p in venv.ps)


def test_qmix_update_launch_grouping(monkeypatch):
    """Host logic of the QMIX update (no GPU: the launch wrappers are replaced by recorders): the agent networks and the
    mixer's hyper-networks share grouped launches -- 3 forward, 2 data-gradient, 1 weight-gradient launch per update --
    and no launch exceeds the library's 8 groups (csrc/gemm.hip MAX_GROUPS)."""
    import torch
    from xuance_amd import nets, ops
    calls = []
    monkeypatch.setattr(ops, "gemm_desc", lambda *a, **k: (a, k))
    monkeypatch.setattr(ops, "linear_fwd", lambda g: calls.append(("fwd", len(g))))
    monkeypatch.setattr(ops, "linear_bwd_data", lambda g: calls.append(("dg", len(g))))
    monkeypatch.setattr(ops, "linear_bwd_weight", lambda g, s, stride: calls.append(("wg", len(g))))
    m = nets.MixingQNet(3, 30, 9, 48, (64,), (64,), 32, 32, "relu", device="cpu")
    B, R = 32, 96
    X, S = torch.zeros(2 * R, 30), torch.zeros(2 * B, 48)
    outs = nets.Plan.forward_many([(m.agent_plan, X, 30, 2 * R, None), (m.agent_target_plan, X[R:], 30, R, m.target_flat),
                                   (m.mixer_plan, S, 48, B, None), (m.mixer_target_plan, S[B:], 48, B, m.target_flat)])
    assert [tuple(o.shape) for o in outs[:2]] == [(2 * R, 9), (2 * R, 9)] or outs[0].shape[1] == 9
    nets.Plan.backward_many([(m.agent_plan, X, 30, R), (m.mixer_plan, S, 48, B)], torch.zeros(4, m.params.P), 1)
    assert calls == [("fwd", 6), ("fwd", 8), ("fwd", 2), ("dg", 4), ("dg", 1), ("wg", 8)]
    assert max(n for _, n in calls) <= 8
    # 8 weight-gradient descriptors cover the 3 agent layers + 7 hyper-network layers (the three ReLU first layers of the
    # hyper-networks are one stacked GEMM)
    assert len(m.trainable_order) == 2 * (3 + 7)


def test_oracle_philox_known_answer_vectors():
    """oracle.philox4x32 (the restatement of csrc/rng.h that lets the oracle replay the device's reset states and action
    draws) against the three Philox4x32-10 known-answer vectors published with Random123 (kat_vectors: counter, key -> out)."""
    from oracle import xrl_oracle as o
    kat = [((0, 0, 0, 0), 0, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, 0xffffffffffffffff, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0x299f31d0 << 32) | 0xa4093822,
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(int(x) for x in o.philox4x32(key, *ctr)) == out
    # vectorised over env indices, 24-bit uniforms in [0, 1), float64 reset states in [-0.05, 0.05)
    u = o.action_uniforms(5, 1000, 17)
    assert u.dtype == np.float32 and 0 <= u.min() and u.max() < 1 and abs(float(u.mean()) - 0.5) < 0.05
    s = o.cartpole_reset_state(3, np.arange(64), np.arange(64) % 5)
    assert s.shape == (64, 4) and s.dtype == np.float64 and np.abs(s).max() < 0.05
    assert not np.array_equal(s[0], o.cartpole_reset_state(3, np.arange(1), 1)[0])


def test_shm_multi_agent_vec_env_matches_in_process_stepping():
    """ShmSubprocVecMultiAgentEnv (the multi-agent twin of ShmSubprocVecEnv, reference surface
    subproc_vec_maenv.py:8-170): 6 SMAC-3m-shaped host envs in 3 worker processes, step data through the shared block --
    every return value of reset() / step() (obs dicts, reward / terminated dicts, truncated, infos incl. state, avail_actions,
    agent_mask, episode_step / score and the reset_* entries of finished episodes) and the buf_* attributes equal those of
    DummyVecMultiAgentEnv on identically seeded envs, over several auto-resets; actions as dict lists or as an array."""
    from xuance_amd.envs import ShmSubprocVecMultiAgentEnv, DummyVecMultiAgentEnv, HostSMACLikeEnv
    n = 6
    venv = ShmSubprocVecMultiAgentEnv([HostSMACLikeEnv] * n, env_seed=5, in_series=2, device="cpu")
    ref = DummyVecMultiAgentEnv([HostSMACLikeEnv] * n, env_seed=5)
    assert venv.agents == ref.agents and venv.num_agents == 3 and venv.max_episode_steps == ref.max_episode_steps

    def same(a, b, what):
        if isinstance(a, dict):
            assert set(a) == set(b), what
            for k in a:
                same(a[k], b[k], f"{what}/{k}")
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b), what
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, f"{what}[{i}]")
        else:
            assert np.array_equal(np.asarray(a, np.float64), np.asarray(b, np.float64)), what
    o1, i1 = venv.reset()
    o2, i2 = ref.reset()
    same(o1, o2, "reset obs")
    same([{k: i[k] for k in ("state", "avail_actions")} for i in i1], [{k: i[k] for k in ("state", "avail_actions")} for i in i2], "reset info")
    rng = np.random.default_rng(0)
    ended = 0
    avail = list(ref.buf_avail_actions)                              # what the next action must respect: after an episode
    for t in range(90):                                              # end the POST-RESET mask (info["reset_avail_actions"])
        acts = [{a: int(np.flatnonzero(avail[e][a])[rng.integers(0, int(avail[e][a].sum()))]) for a in ref.agents}
                for e in range(n)]
        send = acts if t % 2 else np.array([[d[a] for a in ref.agents] for d in acts])
        r1, r2 = venv.step(send), ref.step(acts)
        for k, (x, y) in enumerate(zip(r1[:4], r2[:4])):
            same(list(x), list(y), f"step {t} item {k}")
        for e in range(n):
            keys = ["state", "avail_actions", "agent_mask", "episode_step", "episode_score"]
            if "reset_obs" in r2[4][e]:
                keys += ["reset_obs", "reset_avail_actions", "reset_state"]
                ended += 1
            assert ("reset_obs" in r1[4][e]) == ("reset_obs" in r2[4][e])
            same({k: r1[4][e][k] for k in keys}, {k: r2[4][e][k] for k in keys}, f"step {t} info {e}")
        same(venv.buf_state, ref.buf_state, "buf_state")
        same(venv.buf_avail_actions, ref.buf_avail_actions, "buf_avail_actions")
        avail = [r2[4][e].get("reset_avail_actions", ref.buf_avail_actions[e]) for e in range(n)]
    assert ended >= n                                               # every env went through an auto-reset at least once
    venv.close(); ref.close()
    assert venv.closed


def test_weight_images_of_the_one_launch_kernels_cover_every_parameter_once():
    """Host side of xrl_qmix_fused_update / xrl_marl_act_gru (no GPU needed: the layout entry points are plain C): the
    parameter -> image maps built from xrl_qmix_fused_layout / xrl_marl_act_gru_layout place every weight of the networks a
    launch reads at its own image slot (matrix rows padded to pad4(K) + 4 floats), leave the padding zero, and follow the
    parameters on refresh(); the LDS the launches ask for fits the 160 KB of a CU at the default group sizes."""
    import torch
    from xuance_amd import ops
    from xuance_amd.nets import MixingQNet
    torch.manual_seed(0)
    net = MixingQNet(3, 30, 9, 48, (64,), (64,), 32, 32, "relu", device="cpu")
    fs = ops.QmixFusedState(net, True, 0.99, 1)
    mp = fs.map.numpy()
    used = mp[mp >= 0]
    n_par = sum(int(np.prod(s)) for s in net.params.shapes.values())
    assert len(used) == n_par == 17258 and len(np.unique(used)) == n_par            # (17 258 trainable, FF)
    img = fs.img_eval.numpy()
    assert np.array_equal(img[used], net.params.flat.numpy()[mp >= 0])
    pad = np.ones(len(img), bool); pad[used] = False
    assert not img[pad].any()
    w = net.params.view("individual_q_networks.shared.critic_head.q_value.2.weight").numpy()   # [9, 64] -> rows 68 apart
    o = int(mp[net.params.offsets["individual_q_networks.shared.critic_head.q_value.2.weight"]])
    assert np.array_equal(img[o:o + 64], w[0]) and np.array_equal(img[o + 68:o + 68 + 64], w[1])
    net.params.flat.mul_(2.0); net.copy_target(); fs.refresh()
    assert np.array_equal(fs.img_eval.numpy()[used], net.params.flat.numpy()[mp >= 0])
    assert np.array_equal(fs.img_target.numpy()[used], net.target_flat.numpy()[mp >= 0])
    assert 0 < fs.lds_bytes() <= 160 * 1024
    assert ops.QmixFusedState(net, True, 0.99, 16).lds_bytes() > 160 * 1024            # too many transitions per workgroup
    rnn = MixingQNet(3, 30, 9, 48, (), (64,), 32, 32, "relu", use_rnn=True, fc_hidden=(64,), recurrent_hidden=64, device="cpu")
    st = ops.MarlActGruState(rnn)
    mpa = st.map.numpy()
    ua = mpa[mpa >= 0]
    agent_names = [n for n in rnn.params.shapes if n.startswith("individual_q_networks")]
    assert len(ua) == sum(int(np.prod(rnn.params.shapes[n])) for n in agent_names) == 31689 and len(np.unique(ua)) == len(ua)
    assert np.array_equal(st.image.numpy()[ua], rnn.params.flat.numpy()[mpa >= 0]) and st.lds_bytes <= 160 * 1024
    # (round 6) one row per workgroup and every layer <= 64 wide: the INTERLEAVED image of the one-thread-per-output kernel -- element
    # (n, k) of matrix l at w[l] + ((k // 4) * ldw[l] + n) * 4 + k % 4, ldw = outputs padded to 64; more rows per workgroup or
    # lds_staged: row-major rows (element (n, k) at w[l] + n * ldw[l] + k) for the kernel that stages the image in LDS
    from xuance_amd import _lib
    import ctypes as C
    for kw, inter in ((dict(), 1), (dict(rows_per_wg=6), 0), (dict(lds_staged=True), 0)):
        st = ops.MarlActGruState(rnn, **kw)
        im = _lib.QaImage()
        _lib.call("xrl_marl_act_gru_layout", C.byref(st.struct), C.byref(im))
        assert im.interleaved == inter and im.lds_bytes == st.lds_bytes <= 160 * 1024
        m2 = st.map.numpy()
        assert np.array_equal(st.image.numpy()[m2[m2 >= 0]], rnn.params.flat.numpy()[m2 >= 0]) and len(np.unique(m2[m2 >= 0])) == len(ua)
        w_ih = [n for n in agent_names if n.endswith("weight_ih_l0")]
        assert len(w_ih) == 1
        o, W = rnn.params.offsets[w_ih[0]], rnn.params.view(w_ih[0]).numpy()            # [192, 64], layer 1 of the image
        nn, kk = 77, 13
        at = im.w[1] + ((kk // 4) * im.ldw[1] + nn) * 4 + kk % 4 if inter else im.w[1] + nn * im.ldw[1] + kk
        assert int(m2[o + nn * 64 + kk]) == at and st.image.numpy()[at] == W[nn, kk]
        assert im.ldw[1] == (192 if inter else 68)


def test_wide_kernel_class_is_recognised_on_the_host():
    """Which networks xrl_ppo_wide_minibatch / xrl_wide_act_step cover (ops.PpoWideState.eligible, pure host logic): the two-branch
    Gaussian class D-256-256-{A | 1} of configs/ppo/mujoco.yaml with D <= 24, A <= 8 and the activations the kernels are
    instantiated for -- nothing else (shared representation, other widths, categorical heads, sigmoid)."""
    from xuance_amd import ops
    from xuance_amd.nets import ActorCriticNet
    mk = lambda *a, **k: ActorCriticNet(*a, device="cpu", init=False, **k)
    assert ops.PpoWideState.eligible(mk(17, 6, "gaussian", (), (256, 256), (256, 256), "leaky_relu", activation_action="tanh"))
    assert ops.PpoWideState.eligible(mk(24, 8, "gaussian", (), (256, 256), (256, 256), "relu", activation_action=None))
    assert not ops.PpoWideState.eligible(mk(25, 6, "gaussian", (), (256, 256), (256, 256), "relu", activation_action="tanh"))
    assert not ops.PpoWideState.eligible(mk(17, 9, "gaussian", (), (256, 256), (256, 256), "relu", activation_action="tanh"))
    assert not ops.PpoWideState.eligible(mk(17, 6, "gaussian", (), (128, 128), (128, 128), "relu", activation_action="tanh"))
    assert not ops.PpoWideState.eligible(mk(17, 6, "gaussian", (256,), (256,), (256,), "relu", activation_action="tanh"))
    assert not ops.PpoWideState.eligible(mk(17, 6, "categorical", (), (256, 256), (256, 256), "relu"))
    assert not ops.PpoWideState.eligible(mk(17, 6, "gaussian", (), (256, 256), (256, 256), "sigmoid", activation_action="tanh"))


def test_epsilon_schedule_from_a_step_counter_equals_the_reference_loop():
    """xrl_dqn_act_tail_t.eps_sched (csrc/td.hip): epsilon of vector step k = (float32)(start - (double)(min(k, k*) * n) * delta) must be
    the value OffPolicyAgent._update_explore_factor (off_policy.py:119-127) has reached after k vector steps -- `e = start -
    current_step * delta` re-evaluated while the PREVIOUS value is above end_greedy, current_step growing by n per step -- for
    every k, including the steps after the schedule has stopped (it stops one step LATE, on the first value <= end, not at end)."""
    import struct
    f32 = lambda x: struct.unpack("f", struct.pack("f", x))[0]
    for n, start, end, decay in ((64, 0.5, 0.05, 10 ** 6), (16, 0.5, 0.05, 16 * 16 * 30), (8, 1.0, 0.01, 1000), (3, 1.0, 0.1, 50),
                                 (5, 0.3, 0.3, 100)):
        delta = (start - end) / (decay / n)
        # the reference's loop
        ref, e, cs = [], start, 0
        for k in range(400):
            ref.append(e)
            cs += n
            if e > end:
                e = start - cs * delta
        # the host constant k* and the launch's arithmetic
        from xuance_amd.agents.dqn_agent import eps_kstar
        kstar = eps_kstar(start, end, delta, n)
        for k in range(400):
            got = f32(start - float(min(k, kstar) * n) * delta)
            assert got == f32(ref[k]), (n, k, got, ref[k])


def test_ring_cursor_of_a_captured_store_equals_the_host_store_sequence():
    """xrl_soa_store_step_ring: slot = (slot_bias + c) mod n_size, filled = min(size_bias + c + 1, n_size) with (slot_bias, size_bias)
    = HipOffPolicyBuffer.ring_bias(counter value) taken once -- against store()'s own ptr / size sequence, across the ring's
    wrap-around and for a capture taken at any point of the filling (the full ring's bias is one constant)."""
    class Ring:                                                # the host mirrors of HipOffPolicyBuffer (memory.py), nothing else
        def __init__(self, n_size):
            self.n_size, self.ptr, self.size = n_size, 0, 0
    from xuance_amd.memory import HipOffPolicyBuffer
    for n_size in (1, 5, 24):
        for start in (0, 3, n_size - 1, n_size, 2 * n_size + 1):
            r = Ring(n_size)
            for _ in range(start):                              # stores before the capture
                r.ptr, r.size = (r.ptr + 1) % n_size, min(r.size + 1, n_size)
            c0 = 1000 + start                                   # the device counter's value at capture (any offset from the step count)
            sb, zb = HipOffPolicyBuffer.ring_bias(r, c0)
            for j in range(3 * n_size + 2):                     # replays: counter c0 + j
                c = c0 + j
                slot, filled = (sb + c) % n_size, min(zb + c + 1, n_size)
                assert slot == r.ptr and filled == min(r.size + 1, n_size), (n_size, start, j)
                r.ptr, r.size = (r.ptr + 1) % n_size, min(r.size + 1, n_size)
                if r.size == n_size:                            # (what DQN_Agent._run_pair does: a new key once the ring is full)
                    sb2, zb2 = HipOffPolicyBuffer.ring_bias(r, c + 1)
                    assert (sb2 - sb) % n_size == 0 and min(zb2 + c + 2, n_size) == n_size


def test_recorded_vec_envs_play_a_tape_back():
    """envs/recorded.py on CPU tensors: step k of the tape comes back through the providers' surface -- eager (host step index) and with
    the device counter + static offsets a captured rollout uses --, the next acted-on observation is reset_obs where the env restarted
    (terminated | truncated, or the caller's `restart` mask: the loops' Atari mode), the multi-agent twin loads its recorded resets in order."""
    import numpy as np
    import torch
    from conftest import load_golden
    from xuance_amd.envs import RecordedVecEnv, RecordedMultiAgentVecEnv
    g = load_golden("agent_dqn")
    S, n = g["step/acts"].shape
    done = g["step/terminals"] | g["step/truncations"]
    cur = np.where(done[:, :, None], g["step/reset_obs"], g["step/next_obs"])
    env = RecordedVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"], g["step/reset_obs"], device="cpu")
    env.reset()
    assert np.array_equal(env.buf_obs.numpy(), g["raw_obs0"])
    for k in range(5):
        env.step_device()
        assert np.array_equal(env.next_obs.numpy(), g["step/next_obs"][k]) and np.array_equal(env.buf_obs.numpy(), cur[k])
        assert np.array_equal(env.reward.numpy(), g["step/rewards"][k]) and np.array_equal(env.terminated.numpy() > 0, g["step/terminals"][k])
    env.reset()
    env.prepare(4)
    for base in (0, 4):                                    # two "rollouts" of 4 steps on the counter + offset form
        for t in range(4):
            env.step_device(offset=t)
            assert np.array_equal(env.next_obs.numpy(), g["step/next_obs"][base + t]) and np.array_equal(env.buf_obs.numpy(), cur[base + t])
        env.advance(4)
    only_trunc = RecordedVecEnv(g["raw_obs0"], g["step/next_obs"], g["step/rewards"], g["step/terminals"], g["step/truncations"],
                                g["step/reset_obs"], device="cpu", restart=g["step/truncations"])
    only_trunc.reset()
    k = int(np.flatnonzero((g["step/terminals"] & ~g["step/truncations"]).any(1))[0])      # a step where an env terminated without truncation
    for _ in range(k + 1):
        only_trunc.step_device()
    e = int(np.flatnonzero(g["step/terminals"][k] & ~g["step/truncations"][k])[0])
    assert np.array_equal(only_trunc.buf_obs.numpy()[e], g["step/next_obs"][k][e])          # it keeps acting on its next observation
    q = load_golden("agent_qmix_rnn")
    resets = [dict(obs=q[f"reset{i}/obs"], state=q[f"reset{i}/state"], avail=q[f"reset{i}/avail"], at=int(q[f"reset{i}/at"])) for i in range(int(q["n_resets"]))]
    ma = RecordedMultiAgentVecEnv(resets, q["step/next_obs"], q["step/next_state"], q["step/next_avail"], q["step/rewards"], q["step/terminals"],
                                  q["step/truncations"], q["step/agent_mask"], q["step/reset_obs"], q["step/reset_state"], q["step/reset_avail"],
                                  q["step/episode_step"], device="cpu")
    pos = 0
    for i, r in enumerate(resets[:3]):
        ma.reset()
        assert np.array_equal(ma.buf_obs.numpy(), r["obs"]) and not ma.steps.any()
        end = resets[i + 1]["at"]
        while pos < end:
            assert np.array_equal(ma.buf_obs.numpy(), q["step/acted_obs"][pos])              # what the reference's loop acted on at this step
            assert np.array_equal(ma.steps.numpy(), q["step/episode_step"][pos] - 1)
            ma.step_device()
            assert np.array_equal(ma.done.numpy() > 0, q["step/done"][pos]) and np.array_equal(ma.end_step.numpy()[q["step/done"][pos]], q["step/episode_step"][pos][q["step/done"][pos]])
            pos += 1
    with pytest.raises(AssertionError):
        ma.step_device(); ma.reset(); ma.reset()                                             # a reset the tape does not hold at this position


def test_epsilon_schedule_constants():
    """agents/dqn_agent.py: eps_kstar (closed form) = the first vector step at which OffPolicyAgent._update_explore_factor stops
    (off_policy.py:119-127), against the loop it replaces; degenerate schedules end instead of spinning."""
    import random
    from xuance_amd.agents.dqn_agent import eps_kstar

    def by_loop(s, e, d, n):
        k, x = 0, s
        while x > e:
            k += 1
            x = s - (k * n) * d
        return k
    rnd = random.Random(0)
    for _ in range(500):
        s = rnd.uniform(0.1, 1.0); e = rnd.uniform(0.0, s); n = rnd.choice([1, 4, 8, 64, 256]); dec = rnd.choice([1000, 5000, 77777, 10 ** 5])
        d = (s - e) / (dec / n)
        assert eps_kstar(s, e, d, n) == by_loop(s, e, d, n), (s, e, d, n)
    assert eps_kstar(0.5, 0.1, 0.0, 8) == 0xffffffff and eps_kstar(0.5, 0.1, float("nan"), 8) == 0xffffffff and eps_kstar(0.5, 0.5, 0.1, 8) == 0
