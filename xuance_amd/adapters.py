"""The learner-level seam: ``REGISTRY_Learners[config.learner](config, policy, callback)`` is called by the reference with
the policy object IT built (xuance/torch/agents/base/agent.py:340-341; MARL: ``(config, agent_grouping, policy, callback)``,
multi_agent_rl/qmix_agents.py:45) -- an ``nn.Module`` such as SharedActorCritic, DeepQNetwork, DuelingDeepQNetwork,
VanillaPolicyGradient or MixingQNetwork.  ``adopt()`` turns such a module (or a bare state_dict of it) into the matching
``xuance_amd.nets`` container:

  * the architecture is read off the parameter names and shapes of the state_dict (the names ARE the reference's:
    nets.* export them already), activations off the module tree (falling back to ``config.activation`` /
    ``config.activation_action``);
  * the values are copied into the flat device buffers, and then every ``nn.Parameter`` of the module is re-pointed at
    its view of those buffers (eval parameters into ``params.flat``, ``target_*`` parameters into ``target_flat``).

After that the module and the HIP engine share storage: ``policy.state_dict()`` / ``policy.parameters()`` are live, the
reference agent's own acting code (``self.policy(obs)``) sees every update the kernels make, and ``save_model`` of either
side writes the same numbers.  Nothing is copied per update.
"""
from collections import OrderedDict

import torch

from . import nets

_ACT = {"LeakyReLU": "leaky_relu", "ReLU": "relu", "Tanh": "tanh", "Sigmoid": "sigmoid"}


class AdoptError(ValueError):
    pass


def is_native(model):
    return hasattr(model, "params") and isinstance(getattr(model, "params"), nets.FlatParams)


def _activation_of(module, default):
    """First activation module found in `module` (mlp_block order: Linear, [norm], activation; layers.py:16-33)."""
    if module is not None and hasattr(module, "modules"):
        for m in module.modules():
            name = type(m).__name__
            if name in _ACT:
                return _ACT[name]
    return default


def _last_activation_of(seq, default):
    """Activation that follows the LAST Linear of an nn.Sequential head (GaussianActorHead: activation_action), or None."""
    if seq is None or not hasattr(seq, "children"):
        return default
    kids = list(seq.children())
    if kids and type(kids[-1]).__name__ in _ACT:
        return _ACT[type(kids[-1]).__name__]
    if not any(type(k).__name__ == "Linear" for k in kids):          # not a real layer stack (e.g. a bare parameter tree)
        return default
    return None


def _chain(sd, prefix):
    """[(out, in)] of ``<prefix>.<2i>.weight`` for i = 0, 1, ... (an nn.Sequential of mlp_blocks without normalisation)."""
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in sd if k.startswith(prefix + ".") and k.endswith(".weight")})
    if idx != list(range(0, 2 * len(idx), 2)):
        raise AdoptError(f"{prefix}: layers {idx} are not a plain Linear(+activation) stack (normalisation layers are "
                         "not supported by the HIP engine)")
    return [tuple(sd[f"{prefix}.{i}.weight"].shape) for i in idx]


def _sub(module, path):
    for p in path.split("."):
        if module is None:
            return None
        module = getattr(module, p, None) if not p.isdigit() else (module[int(p)] if hasattr(module, "__getitem__") else None)
    return module


def _cfg(config, name, default=None):
    return getattr(config, name, default) if config is not None else default


# ------------------------------------------------------------------------------------------------ per architecture
def _actor_critic(sd, module, config, device):
    dist = "categorical" if any(k.startswith("actor.logits.") for k in sd) else "gaussian"
    akey = "actor.logits" if dist == "categorical" else "actor.mu"
    rep_w = sorted((k for k in sd if k.startswith("representation.model.") and k.endswith(".weight")), key=lambda k: int(k.split(".")[2]))
    if any(sd[k].dim() == 4 for k in rep_w):
        # AC_CNN_Atari (cnn.py:53-102, configs/ppo/atari.yaml): Conv2d + ReLU blocks at model.0, .2, ..., nn.Flatten(), then
        # Linear + ReLU blocks; a CategoricalActorHead and a ValueHead on the embedding
        if dist != "categorical":
            raise AdoptError("convolutional SharedActorCritic: only the categorical head (policy Categorical_AC) is built")
        convs, fcs = [k for k in rep_w if sd[k].dim() == 4], [k for k in rep_w if sd[k].dim() == 2]
        nc = len(convs)
        if [int(k.split(".")[2]) for k in convs] != list(range(0, 2 * nc, 2)) or \
                [int(k.split(".")[2]) for k in fcs] != list(range(2 * nc + 1, 2 * nc + 1 + 2 * len(fcs), 2)) or not fcs:
            raise AdoptError("convolutional representation is not Conv2d+ReLU blocks, Flatten, Linear+ReLU blocks (AC_CNN_Atari); "
                             "Basic_CNN's global max-pool in front of an actor-critic is not built")
        rep_mod = _sub(module, "representation")
        shape = getattr(rep_mod, "input_shape", None)                 # (C, H, W) (cnn.py:66); frames are H x W x C
        shape = (shape[1], shape[2], shape[0]) if shape is not None else None
        if shape is None and _cfg(config, "observation_space") is not None:
            shape = tuple(config.observation_space.shape)
        strides = getattr(rep_mod, "strides", None) or _cfg(config, "strides")
        if shape is None or strides is None:
            raise AdoptError("convolutional SharedActorCritic: the frame shape and the strides are not in the state_dict -- pass the "
                             "module (AC_CNN_Atari carries input_shape / strides) or a config with observation_space / strides")
        a, c = _chain(sd, akey), _chain(sd, "critic.values")
        net = nets.ActorCriticCNN(tuple(int(x) for x in shape), a[-1][0], tuple(int(sd[k].shape[2]) for k in convs),
                                  tuple(int(x) for x in strides), tuple(int(sd[k].shape[0]) for k in convs),
                                  tuple(int(sd[k].shape[0]) for k in fcs), tuple(x[0] for x in a[:-1]), tuple(x[0] for x in c[:-1]),
                                  _activation_of(rep_mod, _cfg(config, "activation", "relu")), device=device, init=False)
        if net.n_flat != int(sd[fcs[0]].shape[1]):
            raise AdoptError(f"convolutional SharedActorCritic: {net.n_flat} flattened features for the stated frame shape / strides, "
                             f"the first dense layer takes {int(sd[fcs[0]].shape[1])}")
        return net
    rep = _chain(sd, "representation.model") if rep_w else []
    a, c = _chain(sd, akey), _chain(sd, "critic.values")
    obs_dim = rep[0][1] if rep else a[0][1]
    act = _activation_of(_sub(module, "representation") if rep else _sub(module, "actor"), _cfg(config, "activation", "leaky_relu"))
    aa = None
    if dist == "gaussian":
        aa = _last_activation_of(_sub(module, "actor.mu"), _cfg(config, "activation_action", "tanh"))
    return nets.ActorCriticNet(obs_dim, a[-1][0], dist, [s[0] for s in rep], [s[0] for s in a[:-1]], [s[0] for s in c[:-1]],
                               act, aa, device=device, init=False)


def _actor_only(sd, module, config, device):
    dist = "categorical" if any(k.startswith("actor.actor_head.logits.") for k in sd) else "gaussian"
    key = "actor.actor_head.logits" if dist == "categorical" else "actor.actor_head.mu"
    rep = _chain(sd, "actor.representation.model") if any(k.startswith("actor.representation.model.") for k in sd) else []
    a = _chain(sd, key)
    obs_dim = rep[0][1] if rep else a[0][1]
    act = _activation_of(_sub(module, "actor"), _cfg(config, "activation", "leaky_relu"))
    aa = _last_activation_of(_sub(module, "actor.actor_head.mu"), _cfg(config, "activation_action", "tanh")) if dist == "gaussian" else None
    return nets.ActorNet(obs_dim, a[-1][0], dist, [s[0] for s in rep], [s[0] for s in a[:-1]], act, aa, device=device, init=False)


def _deep_q(sd, module, config, device):
    dueling = any(k.startswith("eval_Q_head.v_model.") for k in sd)
    rep_keys = [k for k in sd if k.startswith("representation.model.") and k.endswith(".weight")]
    conv = any(sd[k].dim() == 4 for k in rep_keys)
    act = _activation_of(_sub(module, "eval_Q_head"), _cfg(config, "activation", "relu"))
    if conv:
        ws = [sd[k] for k in sorted(rep_keys, key=lambda k: int(k.split(".")[2]))]
        rep_mod = _sub(module, "representation")
        shape = getattr(rep_mod, "input_shape", None)                 # Basic_CNN keeps (C, H, W) (cnn.py:24); frames are H x W x C
        shape = (shape[1], shape[2], shape[0]) if shape is not None else None
        if shape is None and _cfg(config, "observation_space") is not None:
            shape = tuple(config.observation_space.shape)
        strides = getattr(rep_mod, "strides", None) or _cfg(config, "strides")
        if shape is None or strides is None:
            raise AdoptError("convolutional DeepQNetwork: the frame shape and the strides are not in the state_dict -- pass "
                             "the module (Basic_CNN carries input_shape / strides) or a config with observation_space / strides")
        if dueling:                                                   # DuelingQValueHead: streams of half the hidden width (q_head.py:55-62)
            v, a = _chain(sd, "eval_Q_head.v_model"), _chain(sd, "eval_Q_head.a_model")
            n_actions, hidden = a[-1][0], tuple(2 * s[0] for s in v[:-1])
        else:
            q = _chain(sd, "eval_Q_head.q_value")
            n_actions, hidden = q[-1][0], tuple(s[0] for s in q[:-1])
        return nets.DeepQCNN(tuple(shape), n_actions, tuple(int(w.shape[2]) for w in ws), tuple(int(s) for s in strides),
                             tuple(int(w.shape[0]) for w in ws), hidden, act, device=device, init=False, dueling=dueling)
    rep = _chain(sd, "representation.model") if rep_keys else []
    if dueling:
        v, a = _chain(sd, "eval_Q_head.v_model"), _chain(sd, "eval_Q_head.a_model")
        obs_dim = rep[0][1] if rep else v[0][1]
        return nets.DeepQNet(obs_dim, a[-1][0], [s[0] for s in rep], [2 * s[0] for s in v[:-1]], act, device=device,
                             init=False, dueling=True)
    q = _chain(sd, "eval_Q_head.q_value")
    obs_dim = rep[0][1] if rep else q[0][1]
    return nets.DeepQNet(obs_dim, q[-1][0], [s[0] for s in rep], [s[0] for s in q[:-1]], act, device=device, init=False)


def _mixing_q(sd, module, config, device, n_agents=None, mixer=None):
    groups = sorted({k.split(".")[1] for k in sd if k.startswith("individual_q_networks.")})
    if len(groups) != 1:
        raise AdoptError(f"MixingQNetwork with agent groups {groups}: the HIP QMIX path is built for parameter sharing "
                         "(one group, agents_marl.py:210-215)")
    g = groups[0]
    pe = f"individual_q_networks.{g}"
    rp = f"{pe}.representation.obs_representation"
    use_rnn = f"{rp}.rnn.weight_ih_l0" in sd
    q = _chain(sd, f"{pe}.critic_head.q_value")
    act = _activation_of(_sub(module, f"individual_q_networks.{g}.critic_head") if module is not None else None,
                         _cfg(config, "activation", "relu"))
    has_mixer = any(k.startswith("eval_Qtot.") for k in sd)
    if mixer is None:
        cls = type(_sub(module, "eval_Qtot")).__name__ if module is not None else ""
        mixer = "QMIX" if has_mixer else ("VDN" if "VDN" in cls else ("Independent" if "Independent" in cls else None))
        if mixer is None:
            raise AdoptError("parameter-free mixer: say which one (VDN_Learner / IQL_Learner pass it)")
    kw = dict(mixer=mixer)
    if has_mixer:
        HH, S = sd["eval_Qtot.hyper_w_1.0.weight"].shape
        H = sd["eval_Qtot.hyper_b_1.weight"].shape[0]
        n_agents = sd["eval_Qtot.hyper_w_1.2.weight"].shape[0] // H
        kw.update(mixer_hidden=int(H), hyper_hidden=int(HH))
    else:
        S = _cfg(config, "state_dim", None) or 1
    if n_agents is None:
        raise AdoptError("number of agents unknown: no QMIX mixer in the state_dict and no agent keys given")
    if use_rnn:
        fc = _chain(sd, f"{rp}.mlp") if any(k.startswith(f"{rp}.mlp.") for k in sd) else []
        G, feat = sd[f"{rp}.rnn.weight_ih_l0"].shape
        RH = sd[f"{rp}.rnn.weight_hh_l0"].shape[1]
        obs_dim = fc[0][1] if fc else feat
        return nets.MixingQNet(n_agents, obs_dim, q[-1][0], int(S), (), [s[0] for s in q[:-1]], activation=act, group=g,
                               device=device, init=False, use_rnn=True, fc_hidden=[s[0] for s in fc], recurrent_hidden=int(RH),
                               rnn="LSTM" if G == 4 * RH else "GRU", **kw)
    rep = _chain(sd, f"{rp}.model")
    return nets.MixingQNet(n_agents, rep[0][1], q[-1][0], int(S), [s[0] for s in rep], [s[0] for s in q[:-1]], activation=act,
                           group=g, device=device, init=False, **kw)


def build_from_state_dict(sd, module=None, config=None, device="cuda", **hints):
    keys = list(sd)
    if any(k.startswith("individual_q_networks.") for k in keys):
        return _mixing_q(sd, module, config, device, **hints)
    if any(k.startswith("eval_Q_head.") for k in keys):
        return _deep_q(sd, module, config, device)
    if any(k.startswith("actor.actor_head.") for k in keys):
        return _actor_only(sd, module, config, device)
    if any(k.startswith("actor.logits.") or k.startswith("actor.mu.") for k in keys) and any(k.startswith("critic.values.") for k in keys):
        return _actor_critic(sd, module, config, device)
    raise AdoptError("unrecognised policy: the HIP learners take SharedActorCritic, VanillaPolicyGradient, DeepQNetwork, "
                     f"DuelingDeepQNetwork and MixingQNetwork (first keys: {keys[:4]})")


# ------------------------------------------------------------------------------------------------ public entry points
def live_views(net):
    """name -> view into the net's flat buffers, in the reference's state_dict order (no copies)."""
    tk = getattr(net, "_target_key", None)
    out = OrderedDict()
    for k in net.ref_order:
        t = tk(k) if tk is not None else None
        out[k] = net.params.view(t, net.target_flat) if t else net.params.view(k)
    return out


def adopt(model, config=None, device=None, **hints):
    """xuance_amd net for `model`: a native net is returned as is; an nn.Module (or a state_dict) of the reference is
    rebuilt on the device and -- for a module -- its parameters are re-pointed at the flat buffers (see module docstring).
    hints: n_agents / mixer for MixingQNetworks whose mixer has no parameters."""
    if is_native(model):
        return model
    module = model if hasattr(model, "state_dict") and hasattr(model, "named_parameters") else None
    sd = module.state_dict() if module is not None else model
    if not isinstance(sd, dict) or not sd:
        raise AdoptError(f"cannot adopt {type(model).__name__}: neither a xuance_amd net, an nn.Module nor a state_dict")
    if device is None:
        device = _cfg(config, "device", "cuda")
        device = "cuda" if str(device).startswith(("cuda", "gpu")) or isinstance(device, int) else str(device)
    sd = OrderedDict((k, (v.detach() if hasattr(v, "detach") else torch.as_tensor(v))) for k, v in sd.items())
    net = build_from_state_dict(sd, module, config, device, **hints)
    if list(net.ref_order) != [k for k in sd if k in set(net.ref_order)] or set(net.ref_order) != set(sd):
        extra, missing = set(sd) - set(net.ref_order), set(net.ref_order) - set(sd)
        raise AdoptError(f"state_dict does not match the rebuilt architecture (unexpected {sorted(extra)[:4]}, missing "
                         f"{sorted(missing)[:4]})")
    net.load_state_dict(sd)
    if module is not None:
        views = live_views(net)
        with torch.no_grad():
            for name, p in module.named_parameters():
                p.data = views[name]                      # same shape / dtype; the storage is now the flat device buffer
        net.module = module                               # what callbacks receive as `policy`
        _hook_writes(module, net)
    return net


def _hook_writes(module, net):
    """The module and the engine share storage, so whatever writes parameters THROUGH the module (policy.load_state_dict -- an
    in-place copy into the views --, policy.copy_target(), a caller's own optimiser) also changes what the engine's derived
    weight images (the one-launch QMIX update's, the acting launches') were built from.  load_state_dict and copy_target -- the
    two writers the reference itself has (drl_learner.py:119-121, value_factorization.py:169-174, deep_q_network.py:95-99) -- bump
    the net's `version` (their owners rebuild on the next use); anything else must call net.touched() itself."""
    touched = getattr(net, "touched", None)
    if touched is None:
        def touched():
            net.version = getattr(net, "version", 0) + 1
    if hasattr(module, "register_load_state_dict_post_hook"):
        module.register_load_state_dict_post_hook(lambda m, incompatible: touched())
    ct = getattr(module, "copy_target", None)
    if callable(ct):
        def copy_target(*a, **k):
            out = ct(*a, **k)
            touched()
            return out
        module.copy_target = copy_target
