"""GPU parity of the host-side PPO path (HipOnPolicyBuffer + ActorCriticNet + PPO_Learner) against fixtures
generated from the unmodified reference (tests/golden/ppo_*.npz, onpolicy_buffer.npz)."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import load_golden, sub, assert_close, EngineFixtureCheck

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


# The one gradient tensor of the C2 fixture (8 192 rows) the engine does not hold at 1e-5 of its own scale: actor.logits.0.bias,
# 1.26e-5 from the reference AND from its float64 twin on every path (layered, role-split, any-shape).  Its 128 entries are sums
# over 8 192 rows that cancel to 0.7 % of sum|terms|; the per-row softmax terms come from expf / logf, whose sub-ulp errors are
# one-sided and do not cancel with the signal (torch's vectorised exp is unbiased; the reference sits 1.6e-7 from its twin).
# Every other tensor of that fixture is within 1e-5 of the float32 reference or -- the critic's, whose float32 sgemm sums are
# 1.5e-4 off in the reference itself -- within 1.1e-7 of the float64 twin (profiles/r03_parity_errors_gpu.json).
# A NAMED exception: every other tensor of the C2 fixture is held at 1e-5.
C2_EXCEPT = {"actor.logits.0.bias": 2e-5}


class Capture:
    def __init__(self):
        self.records = []

    def on_update_start(self, iterations, **kw):
        return {}

    def on_update_end(self, iterations, **kw):
        self.records.append({k: (v.detach().cpu().numpy().copy() if isinstance(v, torch.Tensor) else v)
                             for k, v in kw.items() if k not in ("policy", "info")})
        return {}


def make_learner(dist, g, cls_name="PPO_Learner", size=None):
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import REGISTRY_Learners
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    if dist == "categorical":
        net = ActorCriticNet(4, 2, "categorical", (128,), (128,), (128,), "leaky_relu")
        cfg = Namespace(horizon_size=256, n_epochs=8, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.98)
    elif size == "c4":                                             # configs/ppo/mujoco.yaml:8-15
        net = ActorCriticNet(17, 6, "gaussian", (), (256, 256), (256, 256), "leaky_relu", activation_action="tanh")
        cfg = Namespace(horizon_size=256, n_epochs=16, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.99)
    else:
        net = ActorCriticNet(17, 6, "gaussian", (), (64, 64), (64, 64), "relu", activation_action="tanh")
        cfg = Namespace(horizon_size=256, n_epochs=16, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.99)
    cfg.__dict__.update(learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip),
                        use_grad_clip=True, grad_clip_norm=float(gclip), end_factor_lr_decay=float(ef),
                        distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    cb = Capture()
    if cls_name == "A2C_Learner":
        cfg.running_steps = int(total)                             # the fixture's LinearLR horizon (a2c_learner.py:21)
    learner = REGISTRY_Learners[cls_name](cfg, net, cb)
    assert learner.total_iters == int(total)
    return net, learner, cb


@pytest.mark.parametrize("dist", ["categorical", "gaussian"])
def test_a2c_learner_vs_reference_fixture(dist):
    """A2C_Learner (first on-policy sibling of SURVEY 8f): xrl_ppo_loss_* in mode 1 + the same backward / clip / Adam
    launches, against the unmodified reference's a2c_learner.py run (tests/golden/a2c_*.npz)."""
    g = load_golden(f"a2c_{dist}")
    net, learner, cb = make_learner(dist, g, "A2C_Learner")
    net.load_state_dict(sub(g, "init"))
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    chk = EngineFixtureCheck(g, net, learner, float(lr), end_factor=float(ef), total_iters=int(total))
    for u in range(3):
        b = sub(g, f"u{u}/batch")
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                              advantages=b["advantages"], batch_size=len(b["obs"]))
        ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
        assert set(info) == set(ref_info)                          # the reference's key spelling ("actor-loss", ...)
        lp_scale = max(1.0, float(np.abs(ref_cb["log_prob"]).max()))
        # -(adv * log_prob).mean() with O(1) normalised advantages: the terms have the log-probabilities' magnitude
        assert_close(info["actor-loss"], ref_info["actor-loss"], 1e-5, "actor-loss", scale=float(np.abs(ref_cb["log_prob"]).mean()))
        for k in ("critic-loss", "entropy", "predict_value"):
            assert_close(info[k], ref_info[k], 1e-5, k)
        assert_close(info["learning_rate"], ref_info["learning_rate"], 1e-9, "lr")
        rec = cb.records[-1]
        assert_close(rec["v_pred"], ref_cb["v_pred"], 1e-5, "v_pred")
        assert_close(rec["log_prob"], ref_cb["log_prob"], 1e-6, "log_prob", scale=lp_scale)
        assert_close(rec["loss"], ref_cb["loss"], 1e-5, "loss", scale=max(abs(float(ref_cb["loss"])), abs(float(ref_cb["c_loss"]))))
        chk.after_update(u)
    chk.finish()


@pytest.mark.parametrize("as_objects", [False, True])
@pytest.mark.parametrize("dist", ["categorical", "gaussian"])
def test_ppokl_learner_vs_reference_fixture(dist, as_objects):
    """PPOKL_Learner (xrl_ppo_loss_t.mode = 3 + xrl_ppokl_adapt) against the reference's ppokl_learner.py run
    (tests/golden/ppokl_*.npz, see oracle/make_golden.py: golden_ppokl): losses, KL, gradients, parameters, Adam moments at
    1e-5 and the coefficient schedule exactly.  old_dist as parameter arrays, or (as_objects) as per-sample objects with the
    attributes of the reference's split distributions (logits / mu, std)."""
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import REGISTRY_Learners
    g = load_golden(f"ppokl_{dist}")
    lr, vf, ent, target_kl, kl_coef, gclip, ef, total = g["cfg"]
    if dist == "categorical":
        net = ActorCriticNet(4, 3, "categorical", (128,), (128,), (128,), "leaky_relu")
    else:
        net = ActorCriticNet(17, 6, "gaussian", (), (64, 64), (64, 64), "leaky_relu", activation_action="tanh")
    cfg = Namespace(horizon_size=256, n_epochs=8, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.98,
                    learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), target_kl=float(target_kl),
                    kl_coef=float(kl_coef), use_grad_clip=True, grad_clip_norm=float(gclip), end_factor_lr_decay=float(ef),
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    cb = Capture()
    learner = REGISTRY_Learners["PPOKL_Learner"](cfg, net, cb)
    assert learner.total_iters == int(total) and list(net.ref_order) == [str(n) for n in g["param_names"]]
    net.load_state_dict(sub(g, "init"))
    chk = EngineFixtureCheck(g, net, learner, float(lr), end_factor=float(ef), total_iters=int(total))
    for u in range(int(g["n_updates"])):
        b = sub(g, f"u{u}/batch")
        if dist == "categorical":
            old = {"logits": b["old_a"]}
            if as_objects:
                old = np.array([Namespace(logits=torch.from_numpy(r[None])) for r in b["old_a"]], dtype=object)
        else:
            old = {"mu": b["old_a"], "std": b["old_b"][0]}
            if as_objects:
                old = np.array([Namespace(logits=None, mu=torch.from_numpy(m), std=torch.from_numpy(b["old_b"][0])) for m in b["old_a"]],
                               dtype=object)
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], advantages=b["advantages"],
                              aux_batch={"old_dist": old}, batch_size=len(b["obs"]))
        ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
        assert set(info) == set(ref_info)
        lp_scale = max(1.0, float(np.abs(ref_cb["log_prob"]).max()))
        # actor loss = -(ratio * adv).mean() + kl_coef * kl: scale = the ratios' mean magnitude (advantages are O(1)); kl is a
        # mean of differences of log-probabilities of magnitude lp_scale (their float32 floor); ratio compared as its logarithm
        assert_close(info["actor-loss"], ref_info["actor-loss"], 1e-5, "actor-loss", scale=float(np.abs(ref_cb["ratio"]).mean()))
        assert_close(info["kl"], ref_info["kl"], 1e-6, "kl", scale=lp_scale)
        for k in ("critic-loss", "entropy", "predict_value"):
            assert_close(info[k], ref_info[k], 1e-5, k)
        assert_close(info["learning_rate"], ref_info["learning_rate"], 1e-9, "lr")
        assert learner.kl_coef == float(g["kl_coef_after"][u])
        rec = cb.records[-1]
        assert_close(rec["v_pred"], ref_cb["v_pred"], 1e-5, "v_pred")
        assert_close(rec["log_prob"], ref_cb["log_prob"], 1e-6, "log_prob", scale=lp_scale)
        assert_close(np.log(rec["ratio"]), np.log(ref_cb["ratio"]), 2e-6, "log ratio", scale=lp_scale)
        assert_close(rec["loss"], ref_cb["loss"], 1e-5, "loss", scale=max(abs(float(ref_cb["loss"])), float(np.abs(ref_cb["ratio"]).mean())))
        chk.after_update(u)
    chk.finish()


@pytest.mark.parametrize("dist,size", [("categorical", None), ("gaussian", None), ("categorical", "c1"),
                                       ("categorical", "c2"), ("gaussian", "c4")])
def test_ppo_learner_vs_reference_fixture(dist, size):
    """size: fixtures at the BASELINE minibatches -- C1 128 and C2 8 192 rows on the CartPole net (34 051 parameters),
    C4 4 096 rows on the HalfCheetah-shape Gaussian net 17-256-256 (142 605 parameters)."""
    g = load_golden(f"ppo_{dist}" + (f"_{size}" if size else ""))
    net, learner, cb = make_learner(dist, g, size=size)
    assert list(net.ref_order) == [str(n) for n in g["param_names"]]      # same state_dict order as the reference
    assert net.params.P >= {None: 0, "c1": 34051, "c2": 34051, "c4": 142605}[size]
    net.load_state_dict(sub(g, "init"))
    nu = int(g.get("n_updates", 3))
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    chk = EngineFixtureCheck(g, net, learner, float(lr), end_factor=float(ef), total_iters=int(total), tol=1e-5, tol_except=C2_EXCEPT if size == "c2" else None)
    for u in range(nu):
        b = sub(g, f"u{u}/batch")
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                              advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]},
                              batch_size=len(b["obs"]))
        ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
        # the actor loss is the mean of surrogate terms that largely cancel (normalised advantages): scale = their mean magnitude
        sur = float(np.abs(ref_cb["surrogate2"]).mean())
        assert_close(info["actor_loss"], ref_info["actor_loss"], 1e-5, "actor_loss", scale=sur)
        for k in ("critic_loss", "entropy", "predict_value", "clip_ratio"):
            assert_close(info[k], ref_info[k], 1e-5, k)
        assert_close(info["learning_rate"], ref_info["learning_rate"], 1e-9, "lr")
        rec = cb.records[-1]
        lp_scale = max(1.0, float(np.abs(ref_cb["log_prob"]).max()))
        assert_close(rec["v_pred"], ref_cb["v_pred"], 1e-5, "v_pred")
        for k in ("log_prob", "ratio", "surrogate1", "surrogate2"):
            assert_close(rec[k], ref_cb[k], 1e-6, k, scale=lp_scale)
        assert_close(rec["loss"], ref_cb["loss"], 1e-5, "loss", scale=max(abs(float(ref_cb["loss"])), sur, abs(float(ref_cb["c_loss"]))))
        # clipped gradients (p.grad after the step), parameter steps, and at the end Adam's moments: conftest.LearnerFixtureCheck
        chk.after_update(u)
    chk.finish()
    assert learner.iterations == nu and learner.scheduler.last_epoch == nu


def _load_rows(memory, b, n, T):
    """Rows of a fixture minibatch into the HBM rollout buffer (time-major fields; the reference's flat sample index is
    env * T + t, memory_tools.py:270): row r of the batch becomes sample index r."""
    f, dev = memory.soa.fields, memory.soa.fields["returns"].device
    tm = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float32).reshape((n, T) + np.asarray(a).shape[1:])
                                                        .swapaxes(0, 1)), device=dev)
    f["observations"].copy_(tm(b["obs"]).reshape(f["observations"].shape))
    f["actions"].copy_(tm(b["actions"]).reshape(f["actions"].shape))
    f["returns"].copy_(tm(b["returns"]))
    f["advantages"].copy_(tm(b["advantages"]))
    f["aux_old_logp"].copy_(tm(b["old_logp"]))


@pytest.mark.parametrize("size,n,T,kernel", [("c2", 32, 256, "split"), ("c1", 4, 32, "split"), ("c1", 4, 32, "unsplit"),
                                             ("c2", 32, 256, "any-shape"), ("c2", 32, 256, "pair"), ("c1", 4, 32, "pair"),
                                             ("c2", 32, 256, "pair-f32"), ("c1", 4, 32, "pair-f32")])
def test_one_launch_minibatch_kernels_vs_reference_fixture(size, n, T, kernel):
    """The ONE-LAUNCH minibatch kernels pinned to the reference directly, at its own sizes: the rows of the C2 (8 192) / C1
    (128) fixtures are loaded into a HipOnPolicyBuffer and go through gather -> forward -> loss -> backward (xrl_ppo_fused_minibatch:
    ppo_trunk_kernel with (32-row tile, role) workgroups -- "split": 512 workgroups / 256 gradient slabs at C2 -- and with (64-row
    tile, role) workgroups -- "pair": 128 slabs at C2, the headline's kernel: ppo_trunk_bx_kernel, the 128-wide products as exact
    3-way bf16 splits on the matrix cores; "pair-f32": the same tiles on the float32 matrix instruction (use_split_products: False)
    --, the any-shape ppo_fused_kernel with the specialised
    kernels switched off or the role split declined -- "unsplit")
    and xrl_reduce_adam, exactly as PPO_Agent's update phase enqueues them; compared with the reference's `u*/grad` (clipped),
    its float64 twin, its parameter steps and Adam moments (reference: ppo_learner.py:46-67).  The fixture's advantages are
    already normalised (what buffer.sample hands the learner), so the launch gets no statistics."""
    from xuance_amd import ops
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    g = load_golden(f"ppo_categorical_{size}")
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128], critic_hidden_size=[128],
                    activation="leaky_relu", seed=1, parallels=n, running_steps=int(total) * n * T, horizon_size=T, n_epochs=1,
                    n_minibatch=1, learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip),
                    gamma=0.98, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True, grad_clip_norm=float(gclip),
                    end_factor_lr_decay=float(ef), use_obsnorm=False, use_rewnorm=False, obsnorm_range=5, rewnorm_range=5,
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_models", use_hip_graph=False,
                    use_role_split_update=(kernel in ("split", "pair", "pair-f32")), use_pair_update=kernel.startswith("pair"),
                    use_split_products=(kernel == "pair"))
    prev = ops.fast_kernels_enabled()
    ops.set_fast_kernels(kernel != "any-shape")
    try:
        agent = PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=1))
        mem, lr_ = agent.memory, agent.learner
        assert lr_.total_iters == int(total) and lr_.fused_eligible(mem) and agent.batch_size == n * T
        agent.model.load_state_dict(sub(g, "init"))
        chk = EngineFixtureCheck(g, agent.model, lr_, float(lr), end_factor=float(ef), total_iters=int(total),
                                 tol=1e-5, tol_except=C2_EXCEPT if size == "c2" else None)
        idx = torch.arange(n * T, dtype=torch.int64, device="cuda").view(1, -1)
        lr_.prepare_fused(mem, n * T)
        assert lr_.split == (kernel in ("split", "pair", "pair-f32")) and lr_.pair == kernel.startswith("pair")
        assert (lr_.frag16 is not None) == (kernel == "pair")
        lr_.prepare_rows(idx.numel())
        for u in range(int(g["n_updates"])):
            _load_rows(mem, sub(g, f"u{u}/batch"), n, T)
            lr_.refresh_fused_params(mem, idx)                     # derived parameter layouts + this batch's packed / gathered records
            lr_.enqueue_minibatch_fused(mem, idx[0], None)
            info = lr_.last_info(n * T)
            ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
            assert_close(info["actor_loss"], ref_info["actor_loss"], 1e-5, "actor_loss", scale=float(np.abs(ref_cb["surrogate2"]).mean()))
            for k in ("critic_loss", "entropy", "predict_value", "clip_ratio"):
                assert_close(info[k], ref_info[k], 1e-5, k)
            chk.after_update(u)
        chk.finish()
    finally:
        ops.set_fast_kernels(prev)


@pytest.mark.parametrize("dist,size,tiles", [("categorical", "acrobot", 32), ("categorical", "lunar", 32), ("gaussian", "pendulum", 32),
                                             ("gaussian", "walker", 32), ("gaussian", "walker", 64), ("categorical", "acrobot", 64),
                                             ("categorical", "mountaincar", 32), ("categorical", "mountaincar", 64),
                                             ("categorical", "lunar", 64), ("categorical", "acrobot", "64-f32"),
                                             ("categorical", "mountaincar", "64-f32"), ("categorical", "lunar", "64-f32"),
                                             ("gaussian", "pendulum", 64), ("gaussian", "pendulum", "64-f32")])
def test_shared_trunk_family_vs_reference_fixture(dist, size, tiles):
    """The other members of the reference's shared-trunk PPO family (Basic_MLP [128] + actor [128] + critic [128]:
    configs/ppo/classic_control/{Acrobot,Pendulum,MountainCar}.yaml, box2d/{LunarLander,BipedalWalker}.yaml -- (D, A) = (6, 3), (3, 1),
    (2, 3), (8, 4), (24, 4), categorical and Gaussian with tanh on the mean) through the ONE-LAUNCH minibatch kernel (csrc/ppo_trunk.hip: (tile, role)
    workgroups, 32- and 64-row tiles) + xrl_reduce_adam, from rows in a HipOnPolicyBuffer, at the 320-row minibatch their yaml
    gives: the reference learner's loss terms, clipped gradients (float64-anchored), parameter steps, Adam moments.  64-row tiles of
    the members with D <= 8, A <= 4 (categorical, and Pendulum's Gaussian head) run the split-product kernel (csrc/ppo_trunk_bx.hip, any-(D, A) instances); "64-f32":
    the float32 matrix instruction on the same tiles (use_split_products: False)."""
    f32 = tiles == "64-f32"
    tiles = 64 if f32 else tiles
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import PPO_Learner
    from xuance_amd.memory import HipOnPolicyBuffer
    from xuance_amd.spaces import Box, Discrete
    g = load_golden(f"ppo_{dist}_{size}")
    D, A = (int(x) for x in g["shape"])
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    n, T = 10, 32                                                      # 320 rows = one minibatch
    net = ActorCriticNet(D, A, dist, (128,), (128,), (128,), "leaky_relu", activation_action="tanh" if dist == "gaussian" else None)
    assert list(net.ref_order) == [str(x) for x in g["param_names"]]
    net.load_state_dict(sub(g, "init"))
    cfg = Namespace(horizon_size=T, n_epochs=1, n_minibatch=1, parallels=n, running_steps=int(total) * n * T, gamma=0.98,
                    learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip), use_grad_clip=True,
                    grad_clip_norm=float(gclip), end_factor_lr_decay=float(ef), distributed_training=False, device="cuda",
                    model_dir="/tmp/xrl_models", use_pair_update=(tiles == 64), use_split_products=not f32)
    learner = PPO_Learner(cfg, net, Capture())
    assert learner.total_iters == int(total) and learner.trunk_eligible()
    mem = HipOnPolicyBuffer(Box(-np.inf, np.inf, (D,), np.float32), Discrete(A) if dist == "categorical" else Box(-1, 1, (A,), np.float32),
                            {"old_logp": ()}, n, T, device="cuda")
    assert learner.fused_eligible(mem)
    learner.prepare_fused(mem, n * T)
    assert learner.split and learner.pair == (tiles == 64) and learner.params_t is None
    assert (learner.frag16 is not None) == (tiles == 64 and not f32 and D <= 8 and A <= 4)
    chk = EngineFixtureCheck(g, net, learner, float(lr), end_factor=float(ef), total_iters=int(total))
    idx = torch.arange(n * T, dtype=torch.int64, device="cuda").view(1, -1)
    for u in range(int(g["n_updates"])):
        _load_rows(mem, sub(g, f"u{u}/batch"), n, T)
        learner.refresh_fused_params(mem, idx)
        learner.enqueue_minibatch_fused(mem, idx[0], None)
        info = learner.last_info(n * T)
        ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
        assert_close(info["actor_loss"], ref_info["actor_loss"], 1e-5, "actor_loss", scale=float(np.abs(ref_cb["surrogate2"]).mean()))
        for k in ("critic_loss", "entropy", "predict_value", "clip_ratio"):
            assert_close(info[k], ref_info[k], 1e-5, k)
        chk.after_update(u)
    chk.finish()


def ppo_cnn_fc_init(shape):
    """oracle/make_golden.py: ppo_cnn_fc_init (same lines): the dense weight's initial values in the PPO-CNN fixture."""
    i, j = np.meshgrid(np.arange(shape[0], dtype=np.int64), np.arange(shape[1], dtype=np.int64), indexing="ij")
    return (((i * 131 + j * 7919 + 17) % 2003 - 1001).astype(np.float64) * 3e-5).astype(np.float32)


@pytest.mark.parametrize("seam,implicit", [(False, True), (False, False), (True, True)])
def test_ppo_cnn_learner_vs_reference_fixture(seam, implicit):
    """PPO with the convolutional actor-critic of configs/ppo/atari.yaml -- SharedActorCritic(AC_CNN_Atari, CategoricalActorHead,
    ValueHead), the network DummyOnPolicyBuffer_Atari's uint8 frame stacks train (memory_tools.py:290-328) -- through
    PPO_Learner.update on nets.ActorCriticCNN (ConvStack + xrl_flatten_chw_* + the dense plan), against the reference learner's own
    two updates on 32 frames (tests/golden/ppo_cnn_atari.npz, oracle/make_golden.py: golden_ppo_cnn): loss terms, clipped gradients
    of every tensor (the 3.3 M-entry dense weight: the stored rows), parameter steps, Adam moments.  seam: the learner is handed a
    reference-shaped nn.Module (adapters.adopt reads the architecture off its state_dict) instead of the native container.
    implicit: convolutions as implicit GEMMs (csrc/conv_mfma.hip, the default) or im2col + GEMM (csrc/conv.hip)."""
    from xuance_amd.nets import ActorCriticCNN
    from xuance_amd.learners import REGISTRY_Learners
    g = load_golden("ppo_cnn_atari")
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    fc = "representation.model.7.weight"
    init = dict(sub(g, "init"))
    init[fc] = ppo_cnn_fc_init((512, 6400))
    cfg = Namespace(horizon_size=128, n_epochs=4, n_minibatch=4, parallels=8, running_steps=128 * 8 * 100, gamma=0.99,
                    learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip), use_grad_clip=True,
                    grad_clip_norm=float(gclip), end_factor_lr_decay=float(ef), distributed_training=False, device="cuda",
                    model_dir="/tmp/xrl_models", activation="relu", strides=[4, 2, 1],
                    observation_space=Namespace(shape=(84, 84, 4)))
    cb = Capture()
    names = [str(n) for n in g["param_names"]]
    if seam:
        from test_gpu_seam import module_from_state_dict
        module = module_from_state_dict({k: init[k] for k in names})
        learner = REGISTRY_Learners["PPO_Learner"](cfg, module, cb)
        net = learner.model
        assert isinstance(net, ActorCriticCNN) and learner.policy is module
    else:
        net = ActorCriticCNN((84, 84, 4), 4, implicit_conv=implicit)
        net.load_state_dict(init)
        learner = REGISTRY_Learners["PPO_Learner"](cfg, net, cb)
    assert net.conv.implicit == implicit
    assert list(net.ref_order) == names and learner.total_iters == int(total)
    assert sum(int(np.prod(net.params.shapes[k])) for k in names) == 3357861          # conv 77 984 + dense 3 277 312 + heads 2 565
    chk = EngineFixtureCheck(g, net, learner, float(lr), end_factor=float(ef), total_iters=int(total), init_extra={fc: init[fc]},
                             state_source=module if seam else None)
    for u in range(int(g["n_updates"])):
        b = sub(g, f"u{u}/batch")
        assert b["obs"].dtype == np.uint8
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                              advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
        ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
        sur = float(np.abs(ref_cb["surrogate2"]).mean())
        assert_close(info["actor_loss"], ref_info["actor_loss"], 1e-5, "actor_loss", scale=sur)
        for k in ("critic_loss", "entropy", "predict_value", "clip_ratio"):
            assert_close(info[k], ref_info[k], 1e-5, k)
        rec = cb.records[-1]
        assert_close(rec["v_pred"], ref_cb["v_pred"], 1e-5, "v_pred")
        # (1e-5 of the log-probabilities' scale: the logits are sums over 6 400 dense inputs behind three convolutions whose float32
        #  summation order differs from torch's direct convolution -- measured 2.5e-6; the MLP fixtures hold 1e-6)
        for k in ("log_prob", "ratio", "surrogate1", "surrogate2"):
            assert_close(rec[k], ref_cb[k], 1e-5, k, scale=max(1.0, float(np.abs(ref_cb["log_prob"]).max())))
        chk.after_update(u)
    chk.finish()


@pytest.mark.parametrize("size", ["c1", "c2", "c4"])
def test_minibatch_gradient_noise_vs_float64(size):
    """Rounding noise of ONE minibatch gradient at the BASELINE sizes (C1 128 / C2 8 192 rows on the CartPole net, C4 4 096 rows
    on 17-256-256): per tensor, the engine's rms distance from the reference's float64 twin (`u0/grad64`, the reference's learner
    on model.double()) must not exceed the rms distance of the reference's own float32 gradient from it -- the engine's fp32-MFMA
    products are exact fused-multiply-add chains, its slab sums fixed-order, its norm float64 -- beyond 1.5x + 1e-7 of the
    tensor's scale (the twin itself casts the probability ratio to float32, ppo_learner.py:52), or 2e-6 of the tensor's scale
    where the reference is quieter than that (measured: the actor's tensors at C2, 2.4e-7 ... 1.1e-6 against the reference's
    4e-8 ... 1.4e-7 -- one-sided expf / logf rounding, see C2_EXCEPT; the critic's tensors 5e-9 against the reference's 3.5e-6).  Layered path (C1 / C2) and the
    one-launch wide kernel (C4); the CartPole one-launch kernels meet the same twin in
    test_one_launch_minibatch_kernels_vs_reference_fixture."""
    dist = "gaussian" if size == "c4" else "categorical"
    g = load_golden(f"ppo_{dist}_{size}")
    net, learner, cb = make_learner(dist, g, size=size)
    net.load_state_dict(sub(g, "init"))
    b = sub(g, "u0/batch")
    learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"], advantages=b["advantages"],
                   aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
    worse = []
    for k, r32 in sub(g, "u0/grad").items():
        r64 = g[f"u0/grad64/{k}"].astype(np.float64)
        got = net.params.view(k, learner.optimizer.grad).cpu().numpy().astype(np.float64)
        S = float(np.abs(r64).max())
        e_hip, e_ref = float(np.sqrt(np.mean((got - r64) ** 2))) / S, float(np.sqrt(np.mean((r32 - r64) ** 2))) / S
        from conftest import _record
        bound = max(1.5 * e_ref + 1e-7, 2e-6)
        _record(f"gradient rms noise vs f64 twin, {size} {k} [reference float32: {e_ref:.3e}]", e_hip, e_hip, bound, got.size)
        if e_hip > bound:
            worse.append((k, e_hip, e_ref))
    assert not worse, f"noisier than the reference's float32 gradient and than 2e-6 of the tensor's scale: {worse}"


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("M,act,oact", [(96, "leaky_relu", "tanh"), (100, "relu", None), (1000, "tanh", "tanh"), (37, "leaky_relu", "tanh")])
def test_wide_minibatch_kernel_matches_the_layered_path(M, act, oact, split):
    """xrl_ppo_wide_minibatch (one launch: both branches of the 17-256-256-{6 | 1} Gaussian actor-critic forward, loss,
    backward) against the layered path (grouped GEMM launches + xrl_ppo_loss_gaussian) on the same random minibatch, two
    chained updates: ragged last tiles (M % 32 != 0), every activation pair the kernel is instantiated for.  1e-5 on
    everything the reference's update reports (the c4 fixture of test_ppo_learner_vs_reference_fixture pins the same kernel
    to the reference itself).  split: the middle layers' weight gradient as xrl_wide_dw1's launch over all rows (the default) or
    inside the minibatch launch, tile by tile."""
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import REGISTRY_Learners
    rng = np.random.default_rng(M)
    out = {}
    for wide in (True, False):
        torch.manual_seed(5)
        net = ActorCriticNet(17, 6, "gaussian", (), (256, 256), (256, 256), act, activation_action=oact)
        cfg = Namespace(horizon_size=256, n_epochs=16, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.99,
                        learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5,
                        end_factor_lr_decay=0.5, distributed_training=False, device="cuda", model_dir="/tmp/xrl_models",
                        use_fused_update=wide, use_wide_split_dw1=split)
        cb = Capture()
        lr = REGISTRY_Learners["PPO_Learner"](cfg, net, cb)
        assert lr.wide_eligible() == wide
        r = np.random.default_rng(M)
        res = []
        for u in range(2):
            obs = r.standard_normal((M, 17)).astype(np.float32)
            acts = (0.3 * r.standard_normal((M, 6))).astype(np.float32)
            b = dict(obs=obs, actions=acts, returns=r.standard_normal(M).astype(np.float32),
                     values=r.standard_normal(M).astype(np.float32), advantages=r.standard_normal(M).astype(np.float32),
                     aux_batch={"old_logp": (-2.0 + 0.2 * r.standard_normal(M)).astype(np.float32)}, batch_size=M)
            info = lr.update(**b)
            rec = cb.records[-1]
            res.append(dict(info=info, grad=lr.optimizer.grad.cpu().numpy().copy(), params=net.params.flat.cpu().numpy().copy(),
                            **{k: rec[k] for k in ("v_pred", "log_prob", "ratio", "surrogate1", "surrogate2", "a_dist")}))
        out[wide] = res
    for a, b in zip(out[True], out[False]):
        for k in ("actor_loss", "critic_loss", "entropy", "predict_value", "clip_ratio", "learning_rate"):
            assert_close(a["info"][k], b["info"][k], 1e-5, k)
        sc = max(1.0, float(np.abs(b["log_prob"]).max()))
        for k in ("v_pred", "a_dist"):
            assert_close(a[k], b[k], 1e-5, k)
        for k in ("log_prob", "ratio", "surrogate1", "surrogate2"):
            assert_close(a[k], b[k], 1e-5, k, scale=sc * max(1.0, float(np.abs(b[k]).max())))
        assert_close(a["grad"], b["grad"], 1e-5, "clipped gradient")
        assert_close(a["params"], b["params"], 1e-5, "parameters")


@pytest.mark.parametrize("tag", ["gae", "nogae"])
def test_onpolicy_buffer_vs_reference_fixture(tag):
    from xuance_amd.memory import HipOnPolicyBuffer
    from xuance_amd.spaces import Box, Discrete
    g = load_golden("onpolicy_buffer")
    n_envs, T, D, gamma, lam = g["meta"]
    n_envs, T, D = int(n_envs), int(T), int(D)
    d = sub(g, tag)
    buf = HipOnPolicyBuffer(Box(-1, 1, (D,)), Discrete(2), {"old_logp": ()}, n_envs, T, use_gae=(tag == "gae"),
                            use_advnorm=True, gamma=float(gamma), gae_lam=float(lam))
    for t in range(T):
        buf.store(d["obs"][t], d["act"][t], d["rew"][t], d["val"][t], d["term"][t], {"old_logp": d["logp"][t]})
        if buf.full:                                           # ppo_agent.py:129-142
            for i in range(n_envs):
                buf.finish_path(0.0 if d["term"][t, i] else d["boot"][t, i], i)
            returns, advantages = buf.returns.cpu().numpy(), buf.advantages.cpu().numpy()
            s = buf.sample(d["idx"])
            s = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in s.items()}
            s["aux_batch"] = {k: v.cpu().numpy() for k, v in s["aux_batch"].items()}
            buf.clear()
        for i in range(n_envs):                                # ppo_agent.py:146-157
            if d["term"][t, i] or d["trunc"][t, i]:
                buf.finish_path(0.0 if d["term"][t, i] else d["boot"][t, i], i)
    if tag == "gae":
        assert np.array_equal(advantages, d["advantages"]) and np.array_equal(returns, d["returns"])   # bit exact
    else:
        assert_close(advantages, d["advantages"], 1e-6)
        assert_close(returns, d["returns"], 1e-6)
    assert np.array_equal(s["obs"], d["s_obs"]) and np.array_equal(s["actions"], d["s_actions"])
    assert_close(s["returns"], d["s_returns"], 1e-6)
    assert np.array_equal(s["values"], d["s_values"])
    assert np.array_equal(s["aux_batch"]["old_logp"], d["s_old_logp"])
    assert_close(s["advantages"], d["s_advantages"], 1e-6, "adv-norm")
    assert s["batch_size"] == len(d["idx"])
    assert buf.ptr == 0 and buf.size == 0 and float(buf.field("returns").abs().sum()) == 0.0


@pytest.mark.parametrize("tag,dtype", [("f32", "float32"), ("u8", "uint8")])
def test_offpolicy_buffer_vs_reference_fixture(tag, dtype):
    from xuance_amd.memory import HipOffPolicyBuffer, HipOffPolicyBuffer_Atari
    from xuance_amd.spaces import Box, Discrete
    g = load_golden("offpolicy_buffer")
    d = sub(g, tag)
    n_envs, n_size, bs, steps, ptr, size = [int(x) for x in d["meta"]]
    cls = HipOffPolicyBuffer_Atari if tag == "u8" else HipOffPolicyBuffer
    buf = cls(Box(0, 255, d["obs"].shape[2:]), Discrete(4), None, n_envs, n_envs * n_size, bs)
    for t in range(steps):
        buf.store(d["obs"][t], d["act"][t], d["rew"][t], d["term"][t], d["nxt"][t])
    assert (buf.ptr, buf.size) == (ptr, size)
    np.random.seed(123)                                         # same global-RNG draws as memory_tools.py:376-377
    s = buf.sample()
    assert np.array_equal(s["obs"].cpu().numpy(), d["s_obs"])
    assert np.array_equal(s["obs_next"].cpu().numpy(), d["s_obs_next"])
    assert np.array_equal(s["actions"].cpu().numpy(), d["s_actions"].astype(np.float32))
    assert np.array_equal(s["rewards"].cpu().numpy(), d["s_rewards"])
    assert np.array_equal(s["terminals"].cpu().numpy(), d["s_terminals"].astype(np.float32))
    assert str(s["obs"].dtype).endswith(dtype)


def test_resume_from_a_reference_checkpoint():
    """A fresh learner loads the `.pth` the REFERENCE wrote after two updates and makes the third one: parameters equal
    the ones the reference reaches when IT resumes from that file (tests/golden/ppo_ckpt.npz, oracle/make_golden.py)."""
    import os
    from conftest import GOLDEN
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import PPO_Learner
    g = load_golden("ppo_ckpt")
    cfg = Namespace(horizon_size=256, n_epochs=8, n_minibatch=8, parallels=4, running_steps=120000, gamma=0.98,
                    learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5,
                    distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    net = ActorCriticNet(4, 2, "categorical", (128,), (128,), (128,), "leaky_relu")
    learner = PPO_Learner(cfg, net)
    learner.load_model(os.path.join(GOLDEN, "ppo_ckpt_ref.pth"))
    b = sub(g, "u2/batch")
    info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                          advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
    assert_close(info["actor_loss"], g["resumed_info/actor_loss"], 1e-5, "actor_loss")
    assert_close(info["learning_rate"], g["resumed_info/learning_rate"], 1e-9, "lr")
    for k, rp in sub(g, "resumed").items():
        assert_close(net.state_dict()[k].cpu().numpy(), rp, 1e-5, f"param {k} after the resumed update")
    assert learner.optimizer.read().step == 3


def test_resume_continues_the_decayed_learning_rate_like_the_reference():
    """end_factor_lr_decay = 0.5 over 128 iterations: the reference saves after two updates (lr already decayed), a fresh
    reference learner resumes from the file and updates twice more.  Same file, same batches here: the learning rates of
    both resumed updates (1e-9), the parameters (1e-5), and what a re-saved optimizer state reports."""
    import os
    from conftest import GOLDEN
    from xuance_amd.nets import ActorCriticNet
    from xuance_amd.learners import PPO_Learner
    g = load_golden("ppo_ckpt_decay")
    cfg = Namespace(horizon_size=256, n_epochs=8, n_minibatch=8, parallels=4, running_steps=2048, gamma=0.98,
                    learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, use_grad_clip=True, grad_clip_norm=0.5,
                    end_factor_lr_decay=0.5, distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    net = ActorCriticNet(4, 2, "categorical", (128,), (128,), (128,), "leaky_relu")
    learner = PPO_Learner(cfg, net)
    assert learner.total_iters == int(g["total_iters"]) == 128
    learner.load_model(os.path.join(GOLDEN, "ppo_ckpt_decay_ref.pth"))
    assert_close(learner.learning_rate, g["saved_info/learning_rate"], 1e-12, "restored lr")
    assert float(g["saved_info/learning_rate"]) < 4e-4
    for tag, u in (("resumed", 2), ("resumed2", 0)):
        b = sub(g, f"u{u}/batch")
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                              advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
        assert_close(info["learning_rate"], g[f"{tag}_info/learning_rate"], 1e-9, f"lr after the {tag} update")
        for k, rp in sub(g, tag).items():
            assert_close(net.state_dict()[k].cpu().numpy(), rp, 1e-5, f"param {k} ({tag})")
    osd = learner.optimizer.state_dict()["param_groups"][0]
    assert osd["initial_lr"] == 4e-4 and abs(osd["lr"] - float(g["resumed2_info/learning_rate"])) < 1e-12
    assert learner.optimizer.read().step == 4


@pytest.mark.parametrize("dist", ["categorical", "gaussian"])
def test_pg_learner_vs_reference_fixture(dist):
    """PG_Learner on the actor-only model (VanillaPolicyGradient): `xrl_ppo_loss_*` in mode 2 (weight = returns, no critic
    columns) against the reference's pg_learner.py run (tests/golden/pg_*.npz)."""
    from xuance_amd.nets import ActorNet
    from xuance_amd.learners import PG_Learner
    g = load_golden(f"pg_{dist}")
    lr, ent, gclip, ef, total = g["cfg"]
    if dist == "categorical":
        net = ActorNet(4, 2, "categorical", (128,), (128,), "leaky_relu")
    else:
        net = ActorNet(17, 6, "gaussian", (), (64, 64), "relu", activation_action="tanh")
    assert list(net.ref_order) == [str(n) for n in g["param_names"]]
    net.load_state_dict(sub(g, "init"))
    cfg = Namespace(horizon_size=256, n_epochs=1, n_minibatch=1, parallels=4, running_steps=120000, gamma=0.98,
                    learning_rate=float(lr), ent_coef=float(ent), use_grad_clip=True, grad_clip_norm=float(gclip),
                    end_factor_lr_decay=float(ef), distributed_training=False, device="cuda", model_dir="/tmp/xrl_models")
    cb = Capture()
    learner = PG_Learner(cfg, net, cb)
    assert learner.total_iters == int(total)
    # categorical: the head's 2-element bias gradient is +-(one sum over 96 rows cancelling to ~3 % of its terms): per-term
    # float32 rounding shows at 2e-5 of that tensor's scale (measured: 1.9e-5 here, 2.0e-5 in the NumPy oracle) -- THAT tensor is
    # held at 3e-5 (a named exception), every other one at 1e-5
    chk = EngineFixtureCheck(g, net, learner, float(lr), end_factor=float(ef), total_iters=int(total), tol=1e-5,
                             tol_except={"actor.actor_head.logits.2.bias": 3e-5} if dist == "categorical" else None)
    for u in range(3):
        b = sub(g, f"u{u}/batch")
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], batch_size=len(b["obs"]))
        ref_info, ref_cb = sub(g, f"u{u}/info"), sub(g, f"u{u}/cb")
        assert set(info) == set(ref_info)
        # -(returns * log_prob).mean(): O(1) returns times the log-probabilities
        assert_close(info["actor-loss"], ref_info["actor-loss"], 1e-5, "actor-loss", scale=float(np.abs(ref_cb["log_prob"]).mean()))
        assert_close(info["entropy"], ref_info["entropy"], 1e-5, "entropy")
        assert_close(info["learning_rate"], ref_info["learning_rate"], 1e-9, "lr")
        rec = cb.records[-1]
        assert_close(rec["log_prob"], ref_cb["log_prob"], 1e-6, "log_prob", scale=max(1.0, float(np.abs(ref_cb["log_prob"]).max())))
        chk.after_update(u)
    chk.finish()


def _pair_agent(n, T, split_products, seed=3, **kw):
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceCartPoleVecEnv
    cfg = Namespace(representation="Basic_MLP", representation_hidden_size=[128], actor_hidden_size=[128], critic_hidden_size=[128],
                    activation=kw.get("activation", "leaky_relu"), seed=seed, parallels=n, running_steps=1000 * n * T, horizon_size=T,
                    n_epochs=1, n_minibatch=1, learning_rate=7e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, gamma=0.98, use_gae=True,
                    gae_lambda=0.95, use_advnorm=True, use_grad_clip=True, grad_clip_norm=0.5, end_factor_lr_decay=0.5,
                    use_obsnorm=False, use_rewnorm=False, obsnorm_range=5, rewnorm_range=5, distributed_training=False, device="cuda",
                    model_dir="/tmp/xrl_models", use_hip_graph=False, use_role_split_update=True, use_pair_update=True,
                    use_split_products=split_products)
    return PPO_Agent(cfg, DeviceCartPoleVecEnv(n, seed=seed))


def _bf16_planes_to_float(img, plane):
    """float32 value of every element of a three-plane bf16 image: (h + m) + l, each part widened exactly."""
    u = img.view(torch.int16).to(torch.int32) & 0xFFFF
    f = (u << 16).view(torch.float32)
    return (f[:plane] + f[plane:2 * plane]) + f[2 * plane:3 * plane], f


def test_split_fragment_image_is_exact_and_follows_its_maps():
    """xrl_pack_mid_frags16: every weight of the branch layer appears in both sections of the image as three bf16 parts whose sum IS
    the float32 weight (bit for bit), |m| <= 2^-8 |h|, |l| <= 2^-16 |h|; the positions are the ones the split mirror maps name
    (ops.frag16_layout_maps = csrc/split3.h's index formulas); and after optimiser steps through those maps (xrl_reduce_adam's split
    mirror stores) the image equals a fresh pack of the new parameters."""
    from xuance_amd import ops
    agent = _pair_agent(32, 64, True)
    lr_, net = agent.learner, agent.model
    lr_.prepare_fused(agent.memory, 32 * 64)
    assert lr_.frag16 is not None and lr_.pair
    P = net.params.P
    torch.manual_seed(5)
    net.params.flat.copy_(torch.randn(P, device="cuda") * torch.logspace(-6, 2, P, device="cuda")[torch.randperm(P, device="cuda")])
    lr_.refresh_fused_params()
    torch.cuda.synchronize()
    plane = ops.FRAG16_PLANE
    val, parts = _bf16_planes_to_float(lr_.frag16, plane)
    mf, mb = ops.frag16_layout_maps(net.plan, P, "cuda")
    for mp in (mf, mb):
        i = torch.nonzero(mp <= -2).flatten()
        assert i.numel() == 256 * 128
        e = (-(mp[i].to(torch.int64) + 2))
        w = net.params.flat[i]
        assert torch.equal(val[e].view(torch.int32), w.view(torch.int32))
        h, m, l = parts[e], parts[plane + e], parts[2 * plane + e]
        assert bool((m.abs() <= h.abs() * 2.0 ** -8).all()) and bool((l.abs() <= h.abs() * 2.0 ** -16).all())
        # ... and each part is the oracle's (oracle/split3.py: round-to-nearest-even at every step), bit for bit
        from oracle import split3 as S
        oh, om, ol = S.split3(w.cpu().numpy())
        for dev, ref in ((h, oh), (m, om), (l, ol)):
            assert np.array_equal(dev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert torch.unique(torch.cat([-(mf[mf <= -2].long() + 2), -(mb[mb <= -2].long() + 2)])).numel() == plane
    # optimiser steps keep it current
    agent.train(2 * 64)
    torch.cuda.synchronize()
    kept = lr_.frag16.clone()
    lr_.frag16.zero_()
    ops.pack_mid_frags16(net.plan, net.params.flat, lr_.frag16)
    torch.cuda.synchronize()
    assert torch.equal(kept, lr_.frag16)


@pytest.mark.parametrize("activation", ["leaky_relu", "relu", "tanh"])
def test_split_product_minibatch_against_the_float32_instruction(activation):
    """ppo_trunk_bx_kernel beside ppo_trunk_kernel<.., 64, 4, 2> on the same rollout and the same parameters: every gradient slab
    element and loss partial within 2e-6 (tanh: 5e-6) of the tensor's scale (the six-product form drops <= 2^-23 per scalar product;
    everything outside the three products is the same statement), the first-layer products bit-identical -- and the weight-gradient
    operands through the LDS transpose read or through 2-byte reads give the SAME bits (they are two ways of loading the same planes)."""
    from xuance_amd import ops
    n, T = 64, 64
    out = {}
    for tag, split in (("bx", True), ("f32", False)):
        agent = _pair_agent(n, T, split, activation=activation)
        lr_, mem = agent.learner, agent.memory
        agent.train(T)                                                # one rollout + one update phase: a real buffer, moved parameters
        torch.cuda.synchronize()
        if tag == "bx":
            ref_agent = agent
        else:                                                         # same parameters and rollout for the comparison launch
            agent.model.params.flat.copy_(ref_agent.model.params.flat)
            for k, v in ref_agent.memory.soa.fields.items():
                mem.soa.fields[k].copy_(v)
        lr_.prepare_fused(mem, n * T)
        idx = torch.randperm(n * T, device="cuda").view(1, -1)
        if tag != "bx":
            idx = out["idx"]
        out["idx"] = idx
        lr_.prepare_rows(idx.numel())
        lr_.refresh_fused_params(mem, idx)
        assert (lr_.frag16 is not None) == split and lr_.pair
        lr_.enqueue_minibatch_fused(mem, idx[0], None, finish=False)
        torch.cuda.synchronize()
        out[tag] = (lr_.fslabs[:n * T // 64].clone(), lr_.fpartials.clone())
        if tag == "bx":
            ops.set_split_product_tr(False)
            try:
                lr_.fslabs.zero_()
                lr_.enqueue_minibatch_fused(mem, idx[0], None, finish=False)
                torch.cuda.synchronize()
                assert torch.equal(lr_.fslabs[:n * T // 64], out["bx"][0]), "transpose-read and 2-byte-read operands differ"
            finally:
                ops.set_split_product_tr(True)
            for mode in (0, 2):                                       # wave pairs split k in no / both products: other summation orders
                ops.set_split_product_ksplit(mode)
                try:
                    lr_.fslabs.zero_()
                    lr_.enqueue_minibatch_fused(mem, idx[0], None, finish=False)
                    torch.cuda.synchronize()
                    out["bx-ks%d" % mode] = lr_.fslabs[:n * T // 64].clone()
                finally:
                    ops.set_split_product_ksplit(1)
    (sb, pb), (sf, pf) = out["bx"], out["f32"]
    P = ref_agent.model.params.P
    gb, gf = sb.double().sum(0), sf.double().sum(0)
    offs = ref_agent.model.params.offsets
    names = sorted(offs, key=lambda k: offs[k])
    from conftest import _record
    # (tanh: the first layer's gradient multiplies by 1 - h1^2 and sums cancelling rows -- 3.2e-6 measured on its bias; 1.2e-6 is the
    #  worst tensor of the piecewise-linear activations)
    tol = 5e-6 if activation == "tanh" else 2e-6
    for k in names:
        lo, hi = offs[k], offs[k] + int(np.prod(ref_agent.model.params.shapes[k]))
        a, b = gb[lo:hi], gf[lo:hi]
        S = float(b.abs().max()) + 1e-30
        err = float((a - b).abs().max()) / S
        _record(f"split-product minibatch vs float32 instruction, {activation} {k}", err, err, tol, a.numel())
        assert err <= tol, (k, err)
        for mode in (0, 2):
            c = out["bx-ks%d" % mode].double().sum(0)[lo:hi]
            assert float((c - b).abs().max()) / S <= tol, (k, "k-split mode", mode)
    # fold region (the critic role's first-layer gradient) rides behind the parameters
    assert float((gb[P:] - gf[P:]).abs().max()) <= tol * (float(gf[P:].abs().max()) + 1e-30)
    assert torch.allclose(pb, pf, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("records", [True, False])
def test_split_product_minibatch_on_ragged_minibatches_and_field_rows(records):
    """The split-product kernel where its index arithmetic differs from the full-tile record path: a minibatch that ends inside a
    64-row tile (M = 64 k - 17: rows past M must contribute nothing to any sum) and rows gathered through `idx` from the buffer's
    FIELDS instead of the packed 32-byte records.  Against the float32-instruction kernel on the same launch arguments: 2e-6 of each
    tensor's scale, loss partials 1e-6; and the slab rows of tiles behind the last one stay untouched."""
    from xuance_amd import ops
    n, T = 64, 64
    M = n * T - 64 * 5 - 17
    out = {}
    for tag, split in (("bx", True), ("f32", False)):
        agent = _pair_agent(n, T, split, seed=11)
        lr_, mem = agent.learner, agent.memory
        agent.train(T)
        torch.cuda.synchronize()
        if tag == "bx":
            ref_agent = agent
            idx = torch.randperm(n * T, device="cuda")[:M].contiguous()
        else:
            agent.model.params.flat.copy_(ref_agent.model.params.flat)
            for k, v in ref_agent.memory.soa.fields.items():
                mem.soa.fields[k].copy_(v)
        lr_.prepare_fused(mem, n * T)
        lr_.refresh_fused_params(mem, None)                           # packed records of the rollout, no gathered rows
        if not records:
            lr_._packed_valid = False                                 # rows through idx from the fields
        lr_._rows_idx = None
        assert (lr_.frag16 is not None) == split and lr_.pair
        lr_.fslabs.fill_(7.0)
        lr_.enqueue_minibatch_fused(mem, idx, None, finish=False)
        torch.cuda.synchronize()
        nt = (M + 63) // 64
        assert bool((lr_.fslabs[nt:] == 7.0).all()), "a workgroup wrote behind the last tile's slab"
        out[tag] = (lr_.fslabs[:nt].clone(), lr_.fpartials[:2 * nt].clone())
    (sb, pb), (sf, pf) = out["bx"], out["f32"]
    gb, gf = sb.double().sum(0), sf.double().sum(0)
    offs, shapes, P = ref_agent.model.params.offsets, ref_agent.model.params.shapes, ref_agent.model.params.P
    for k in offs:
        lo, hi = offs[k], offs[k] + int(np.prod(shapes[k]))
        S = float(gf[lo:hi].abs().max()) + 1e-30
        err = float((gb[lo:hi] - gf[lo:hi]).abs().max()) / S
        assert err <= 2e-6, (k, err)
    assert float((gb[P:] - gf[P:]).abs().max()) <= 2e-6 * (float(gf[P:].abs().max()) + 1e-30)
    assert torch.allclose(pb, pf, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("dist,D,A,act", [("categorical", 6, 3, "leaky_relu"), ("categorical", 8, 4, "relu"), ("categorical", 2, 3, "tanh"),
                                          ("gaussian", 3, 1, "leaky_relu"), ("categorical", 4, 2, "leaky_relu")])
def test_acting_pass_in_one_launch_matches_the_layered_forward(dist, D, A, act):
    """xrl_trunk_forward16 (the forward-only instances of csrc/ppo_trunk_bx.hip: representation, branch layer as exact 3-way bf16 splits,
    both heads, activation_action -- one launch) against the three launches of the layered forward (nets.Plan.forward: float32 matrix
    instruction) on the same parameters and rows, incl. a last tile that ends inside 64 rows: every logit / mean / value within 2e-6 of
    the head buffer's scale; rows beyond M untouched."""
    from xuance_amd import ops
    from xuance_amd.nets import ActorCriticNet
    torch.manual_seed(3)
    net = ActorCriticNet(D, A, dist, (128,), (128,), (128,), act, activation_action="tanh" if dist == "gaussian" else None)
    net.params.flat.copy_(torch.randn(net.params.P, device="cuda") * 0.2)
    M = 64 * 5 - 23
    X = torch.randn(M + 64, D, device="cuda")
    ref = net.forward(X, M)[:M].clone()
    img = torch.zeros(3 * ops.FRAG16_PLANE, dtype=torch.int16, device="cuda")
    ops.pack_mid_frags16(net.plan, net.params.flat, img)
    out = torch.full((M + 64, A + 1), 7.0, device="cuda")
    gauss = dist == "gaussian"
    ops.trunk_forward16(net.plan, net.params.flat, img, X, M, out, A + 1, D, A, gauss, ops.ACT[net.activation_action] if gauss else 0)
    torch.cuda.synchronize()
    assert bool((out[M:] == 7.0).all())
    S = float(ref.abs().max())
    err = float((out[:M] - ref).abs().max()) / S
    from conftest import _record
    _record(f"one-launch acting pass vs layered forward, {dist} ({D}, {A}) {act}", err, err, 2e-6, ref.numel())
    assert err <= 2e-6, err


def test_general_path_rollout_uses_the_one_launch_acting_pass():
    """PPO on the device Acrobot at the headline's sizes (256 envs, minibatches of 8 192: the update phase runs the split-product kernel,
    so its weight planes exist): the rollout's acting pass goes through xrl_trunk_forward16; with use_trunk_forward: False through the
    layered forward.  Same seeds: the first vector step's values / log-probs agree to 1e-5, its actions are equal, both agents train."""
    import bench
    from xuance_amd.agents import PPO_Agent
    from xuance_amd.envs import DeviceAcrobotVecEnv
    outs = {}
    for tf in (True, False):
        cfg = bench.make_config(256, 256, 1, 0)
        cfg.use_trunk_forward = tf
        torch.manual_seed(1)
        agent = PPO_Agent(cfg, DeviceAcrobotVecEnv(256, seed=1))
        info = agent.train(256)
        torch.cuda.synchronize()
        assert (agent._trunk_forward() is not None) == tf
        f = agent.memory.soa.fields
        assert all(np.isfinite(float(v)) for v in info.values() if isinstance(v, (int, float)))
        agent2 = PPO_Agent(cfg, DeviceAcrobotVecEnv(256, seed=1))     # first step of a fresh agent: same parameters in both modes
        agent2.rollout()
        torch.cuda.synchronize()
        f2 = agent2.memory.soa.fields
        outs[tf] = (f2["values"][0].clone(), f2["aux_old_logp"][0].clone(), f2["actions"][0].clone())
    (v1, l1, a1), (v0, l0, a0) = outs[True], outs[False]
    assert torch.equal(a1, a0)
    assert float((v1 - v0).abs().max()) <= 1e-5 * (float(v0.abs().max()) + 1e-30) and float((l1 - l0).abs().max()) <= 1e-5


@pytest.mark.parametrize("dist,D,A", [("categorical", 6, 3), ("gaussian", 3, 1), ("categorical", 8, 4)])
def test_acting_pass_with_sampling_in_one_launch_matches_forward_plus_policy_sample(dist, D, A):
    """xrl_trunk_forward16 with a sample argument (heads, action sampling, log-prob, values, bootstrap values: one launch, the two heads in
    different workgroups) against layered forward + xrl_policy_sample on the same rows and the same supplied randomness: actions EQUAL,
    log-probs / values / bootstrap values at 1e-5; then the bootstrap-only form (act_out = None) and a Philox-drawn run (same counter:
    same draws, equal actions)."""
    from xuance_amd import ops
    from xuance_amd.nets import ActorCriticNet
    torch.manual_seed(5)
    gauss = dist == "gaussian"
    net = ActorCriticNet(D, A, dist, (128,), (128,), (128,), "leaky_relu", activation_action="tanh" if gauss else None)
    net.params.flat.copy_(torch.randn(net.params.P, device="cuda") * 0.2)
    n = 200
    X = torch.randn(2 * n, D, device="cuda")
    img = torch.zeros(3 * ops.FRAG16_PLANE, dtype=torch.int16, device="cuda")
    ops.pack_mid_frags16(net.plan, net.params.flat, img)
    log_std = net.params.ptr("actor.log_std") if gauss else None
    oact = ops.ACT[net.activation_action] if gauss else 0
    for noise in (torch.rand(n, device="cuda") if not gauss else torch.randn(n, A, device="cuda"), None):
        res = {}
        for tag in ("fused", "layered"):
            act = torch.zeros(n, A if gauss else 1, device="cuda").squeeze(-1) if not gauss else torch.zeros(n, A, device="cuda")
            val, logp, boot = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
            env_a = torch.zeros(n, dtype=torch.int32, device="cuda") if not gauss else torch.zeros(n, A, device="cuda")
            kw = dict(log_std=log_std, noise=noise, act_out=act, val_out=val, logp_out=logp, env_action=None if gauss else env_a,
                      env_action_f=env_a if gauss else None, bootv_prev=boot, n=n, A=A, ld=A + 1, gaussian=int(gauss), seed=77, step=3,
                      step_dev=None)
            if tag == "fused":
                ops.trunk_forward16(net.plan, net.params.flat, img, X, 2 * n, None, 0, D, A, gauss, oact, sample=kw)
            else:
                ops.policy_sample(heads=net.forward(X, 2 * n), **kw)
            torch.cuda.synchronize()
            res[tag] = (act.clone(), val.clone(), logp.clone(), boot.clone(), env_a.clone())
        (a1, v1, l1, b1, e1), (a0, v0, l0, b0, e0) = res["fused"], res["layered"]
        if gauss:
            assert float((a1 - a0).abs().max()) <= 1e-5 * (float(a0.abs().max()) + 1e-30)
        else:
            assert torch.equal(a1, a0) and torch.equal(e1, e0)
        for x, y in ((v1, v0), (l1, l0), (b1, b0)):
            assert float((x - y).abs().max()) <= 1e-5 * max(float(y.abs().max()), 1.0)
    # bootstrap values only
    boot2 = torch.zeros(n, device="cuda")
    ops.trunk_forward16(net.plan, net.params.flat, img, X, 2 * n, None, 0, D, A, gauss, oact,
                        sample=dict(act_out=None, val_out=None, logp_out=None, bootv_prev=boot2, n=n, A=A, ld=A + 1, gaussian=0,
                                    seed=77, step=0, step_dev=None))
    torch.cuda.synchronize()
    assert float((boot2 - res["layered"][3]).abs().max()) <= 1e-5 * max(float(boot2.abs().max()), 1.0)
