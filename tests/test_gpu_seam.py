"""GPU: the learner-level seam on the device.  The HIP learners are constructed the way the reference constructs its
learners -- ``Learner(config, policy_module, callback)`` / ``(config, agent_grouping, policy_module, callback)`` -- with an
``nn.Module`` that carries the reference's parameter names (built here from the fixture state_dicts: the reference itself
does not exist on the GPU box; tests/test_cpu_seam.py does the same with the reference's real classes).  After every
update the MODULE's own state_dict must hold the reference's post-step parameters: module and engine share storage."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import load_golden, sub, assert_close, EngineFixtureCheck

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class _Node(torch.nn.Module):
    pass


def module_from_state_dict(sd):
    """An nn.Module tree whose state_dict() has exactly the keys / order / values of `sd` (CPU tensors, like a freshly
    built reference policy before `.to(device)`)."""
    root = _Node()
    for k, v in sd.items():
        m, parts = root, k.split(".")
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, _Node())
            m = m._modules[p]
        m.register_parameter(parts[-1], torch.nn.Parameter(torch.as_tensor(np.asarray(v)).clone()))
    assert list(root.state_dict()) == list(sd)
    return root


class Capture:
    def __init__(self):
        self.policies = []

    def on_update_start(self, iterations, **kw):
        return {}

    def on_update_end(self, iterations, **kw):
        self.policies.append(kw.get("policy", kw.get("model")))
        return {}


def base(**kw):
    c = dict(distributed_training=False, device="cuda", model_dir="/tmp/xrl_models", running_steps=120000, parallels=4,
             start_training=0, training_frequency=1)
    c.update(kw)
    return Namespace(**c)


def module_check(g, module, learner, lr, end_factor=1.0, total_iters=1):
    """conftest.EngineFixtureCheck reading the parameters through the CALLER's module (they must be live views of the engine's
    flat buffer): gradients at each tensor's scale, steps through Adam's conditioning, target copies exact."""
    for v in module.state_dict().values():
        assert v.is_cuda
    return EngineFixtureCheck(g, learner.model, learner, lr, end_factor=end_factor, total_iters=total_iters, state_source=module)


@pytest.mark.parametrize("name", ["ppo_categorical_c1", "ppo_gaussian_c4"])
def test_ppo_learner_on_a_reference_shaped_module(name):
    from xuance_amd.learners import REGISTRY_Learners
    g = load_golden(name)
    lr, vf, ent, clip, gclip, ef, total = g["cfg"]
    module = module_from_state_dict(sub(g, "init"))
    cfg = base(horizon_size=256, n_epochs=8 if "categorical" in name else 16, n_minibatch=8, gamma=0.98,
               learning_rate=float(lr), vf_coef=float(vf), ent_coef=float(ent), clip_range=float(clip), use_grad_clip=True,
               grad_clip_norm=float(gclip), end_factor_lr_decay=float(ef), activation="leaky_relu", activation_action="tanh")
    cb = Capture()
    learner = REGISTRY_Learners["PPO_Learner"](cfg, module, cb)       # agent.py:340-341
    assert learner.total_iters == int(total) and learner.policy is module
    chk = module_check(g, module, learner, float(lr), float(ef), int(total))
    for u in range(int(g["n_updates"])):
        b = sub(g, f"u{u}/batch")
        info = learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                              advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
        assert_close(info["actor_loss"], sub(g, f"u{u}/info")["actor_loss"], 1e-5, "actor_loss",
                     scale=float(np.abs(sub(g, f"u{u}/cb")["surrogate2"]).mean()))      # a mean of cancelling surrogate terms
        chk.after_update(u)
        assert cb.policies[-1] is module
    # the module's own forward (what the reference agent's acting code runs) works on the shared storage
    p = dict(module.named_parameters())
    key = "actor.logits.0.weight" if "categorical" in name else "actor.mu.0.weight"
    assert p[key].data_ptr() == learner.model.params.view(key).data_ptr()


def test_dqn_learner_on_a_reference_shaped_module():
    from xuance_amd.learners import DQN_Learner
    g = load_golden("dqn_mlp")
    lr, gamma, sync, gclip, use_clip, total = g["cfg"]
    module = module_from_state_dict(sub(g, "init"))
    cb = Capture()
    learner = DQN_Learner(base(learning_rate=float(lr), gamma=float(gamma), sync_frequency=int(sync), use_grad_clip=bool(use_clip),
                               grad_clip_norm=float(gclip), activation="relu"), module, cb)
    chk = module_check(g, module, learner, float(lr), total_iters=int(total))
    for u in range(3):
        b = sub(g, f"u{u}/batch")
        info = learner.update(batch_size=len(b["obs"]), **b)
        assert_close(info["Qloss"], sub(g, f"u{u}/info")["Qloss"], 1e-5, "Qloss")
        chk.after_update(u)                                            # incl. target_* after the hard sync of update 2
    assert cb.policies[-1] is module


def test_qmix_learner_on_a_reference_shaped_module():
    from xuance_amd.learners import QMIX_Learner
    g = load_golden("qmix_ff_double_c5")
    lr, gamma, sync, gclip, dq, total = g["cfg"]
    module = module_from_state_dict(sub(g, "init"))
    keys = [f"agent_{i}" for i in range(3)]
    grouping = Namespace(agent_keys=keys, group_keys=[str(g["group"])])          # AgentGrouping's two attributes
    cb = Capture()
    learner = QMIX_Learner(base(learning_rate=float(lr), gamma=float(gamma), sync_frequency=int(sync), use_grad_clip=True,
                                grad_clip_norm=float(gclip), double_q=bool(dq), use_actions_mask=True, use_parameter_sharing=True,
                                n_epochs=8, activation="relu"), grouping, module, cb)     # qmix_agents.py:45
    assert learner.model.n_agents == 3 and learner.model.state_dim == 48
    chk = module_check(g, module, learner, float(lr), total_iters=int(total))
    for u in range(3):
        b = sub(g, f"u{u}/batch")
        sample = {k: {a: b[k][:, i] for i, a in enumerate(keys)}
                  for k in ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions", "avail_actions_next")}
        sample.update(state=b["state"], state_next=b["state_next"], batch_size=len(b["state"]))
        info = learner.update(sample)
        assert_close(info["loss_Q"], sub(g, f"u{u}/info")["loss_Q"], 1e-5, "loss_Q")
        chk.after_update(u)
    assert cb.policies[-1] is module
