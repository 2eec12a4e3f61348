"""CPU: the learner-level seam (first bullet).  The reference constructs its learners as
``REGISTRY_Learners[config.learner](config, policy, callback)`` with the policy IT built (agent.py:340-341,
qmix_agents.py:45).  Here the reference's own SharedActorCritic / DeepQNetwork (MLP and CNN) / DuelingDeepQNetwork /
VanillaPolicyGradient / MixingQNetwork (feed-forward, GRU, VDN) are built through the import shim (build container only:
/root/reference does not travel) and handed to the HIP learners' constructors on device "cpu" (host-only: construction
touches no kernel): the adapter must rebuild the right architecture, keep the state_dict order and values, and leave the
module's parameters as live views of the flat buffers."""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest

torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/xuance"), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import ref_shim
    ref_shim.install()
    from torch import nn
    from xuance.torch.rl_models.representations import Basic_MLP, Basic_Identical, Basic_CNN, Basic_RNN
    from xuance.torch.rl_models.heads.actor_head import CategoricalActorHead, GaussianActorHead
    from xuance.torch.rl_models.heads.critic_head import ValueHead
    from xuance.torch.rl_models.heads.q_mix_head import QMIX_Mixer
    from xuance.torch.rl_models.heads import VDN_Mixer
    from xuance.torch.rl_models.architectures.single_agent.actor_critic import SharedActorCritic
    from xuance.torch.rl_models.architectures.single_agent.deep_q_network import DeepQNetwork, DuelingDeepQNetwork
    from xuance.torch.rl_models.architectures.multi_agent.value_factorization import MixingQNetwork
    from xuance.torch.rl_models.architectures.single_agent.reinforce import VanillaPolicyGradient
    from xuance.torch.rl_models.actors.categorical_actors import CategoricalActor
    from xuance.torch.rl_models.critics.base_critics import DiscreteActionValueCritic
    from xuance.torch.rl_models.representations.agent_feature import AgentFeatureEncoder
    from xuance.torch.rl_models.modules.identity_encoder import build_identity_encoder, IdentityFeatureFusion
    from xuance.common import AgentGrouping
    return Namespace(**{k: v for k, v in locals().items() if k != "ref_shim"}, sp=ref_shim.spaces())


def cfg(**kw):
    c = dict(distributed_training=False, device="cpu", model_dir="/tmp/xrl_models", running_steps=120000, parallels=4,
             learning_rate=4e-4, gamma=0.99, use_grad_clip=True, grad_clip_norm=0.5, horizon_size=256, n_epochs=8, n_minibatch=8,
             vf_coef=0.25, ent_coef=0.01, clip_range=0.2, sync_frequency=100, start_training=0, training_frequency=1,
             use_parameter_sharing=True, double_q=True, use_actions_mask=True, use_rnn=False, episode_length=60)
    c.update(kw)
    return Namespace(**c)


def check_shared_storage(module, learner):
    net = learner.model
    sd = module.state_dict()
    assert list(sd) == list(net.ref_order)                                # the reference's own key order
    ours = net.state_dict()
    for k, v in sd.items():
        assert v.shape == ours[k].shape and torch.equal(v.cpu(), ours[k].cpu()), k
    # live: what the engine writes is what the module sees (and the other way round), targets in their own buffer
    net.params.flat.add_(1.0)
    if hasattr(net, "target_flat"):
        net.target_flat.sub_(2.0)
    after = module.state_dict()
    for k, v in sd.items():
        d = -2.0 if k.startswith("target_") else 1.0
        assert torch.allclose(after[k], ours[k] + d), k
    n_params = sum(p.numel() for p in module.parameters())
    assert n_params == sum(int(np.prod(net.params.shapes[n])) for n in net.params.names) * (2 if hasattr(net, "target_flat") else 1)
    assert learner.policy is module                                       # callbacks receive the caller's object
    with torch.no_grad():
        for p in module.parameters():
            p.mul_(0.5)                                                   # an in-place edit through the module ...
    assert torch.allclose(net.state_dict()[next(iter(sd))], (ours[next(iter(sd))] + 1.0) * 0.5)   # ... lands in the flat buffer
    # the two writers the reference itself has bump the net's version (derived weight images are rebuilt on their next use)
    v0 = getattr(net, "version", 0)
    module.load_state_dict(module.state_dict())
    assert getattr(net, "version", 0) > v0
    if hasattr(module, "copy_target"):
        v1 = net.version
        module.copy_target()
        assert net.version > v1
        tk = next(k for k in sd if k.startswith("target_"))
        src = tk[len("target_"):] if tk[len("target_"):] in sd else None
        if src is not None:
            assert torch.equal(module.state_dict()[tk], module.state_dict()[src])


def test_ppo_learner_takes_the_references_shared_actor_critic(ref):
    from xuance_amd.learners import PPO_Learner
    init, nn = torch.nn.init.orthogonal_, ref.nn
    rep = ref.Basic_MLP((4,), [128], None, init, nn.LeakyReLU, "cpu")
    model = ref.SharedActorCritic(rep, ref.CategoricalActorHead(128, [128], 2, None, init, nn.LeakyReLU, "cpu"),
                                  ref.ValueHead(128, [128], None, init, nn.LeakyReLU, "cpu"))
    learner = PPO_Learner(cfg(), model, None)
    net = learner.model
    assert (net.obs_dim, net.action_dim, net.dist, net.activation) == (4, 2, "categorical", "leaky_relu")
    assert list(net.plan.widths) == [4, 128, 256, 3] and net.params.P >= 34051
    check_shared_storage(model, learner)
    # Gaussian head of configs/ppo/mujoco.yaml: Basic_Identical, 256-256, tanh on the mean, log_std parameter
    model = ref.SharedActorCritic(ref.Basic_Identical((17,), "cpu"),
                                  ref.GaussianActorHead(17, [256, 256], 6, None, init, nn.ReLU, nn.Tanh, "cpu"),
                                  ref.ValueHead(17, [256, 256], None, init, nn.ReLU, "cpu"))
    learner = PPO_Learner(cfg(activation="leaky_relu"), model, None)      # the module tree wins over the config string
    net = learner.model
    assert (net.dist, net.activation, net.activation_action) == ("gaussian", "relu", "tanh")
    assert sum(p.numel() for p in model.parameters()) == 142605
    check_shared_storage(model, learner)


def test_dqn_learners_take_the_references_networks(ref):
    from xuance_amd.learners import DQN_Learner, DuelDQN_Learner
    init, nn = torch.nn.init.orthogonal_, ref.nn
    model = ref.DeepQNetwork(ref.Basic_MLP((6,), [64], None, init, nn.ReLU, "cpu"), [64], ref.sp.Discrete(4), None, init, nn.ReLU, "cpu")
    learner = DQN_Learner(cfg(), model, None)
    assert list(learner.model.plan.widths) == [6, 64, 64, 4] and not learner.model.dueling
    check_shared_storage(model, learner)
    model = ref.DuelingDeepQNetwork(ref.Basic_MLP((6,), [64], None, init, nn.ReLU, "cpu"), [64], ref.sp.Discrete(4), None, init, nn.ReLU, "cpu")
    learner = DuelDQN_Learner(cfg(), model, None)
    assert learner.model.dueling
    check_shared_storage(model, learner)
    # the Atari network of configs/dqn/atari.yaml
    rep = ref.Basic_CNN((84, 84, 4), [8, 4, 3], [4, 2, 1], [32, 64, 64], None, init, nn.ReLU, "cpu")
    model = ref.DeepQNetwork(rep, [512], ref.sp.Discrete(4), None, init, nn.ReLU, "cpu")
    learner = DQN_Learner(cfg(), model, None)
    net = learner.model
    assert (net.obs_shape, net.kernels, net.strides, net.filters) == ((84, 84, 4), (8, 4, 3), (4, 2, 1), (32, 64, 64))
    assert sum(int(np.prod(net.params.shapes[k])) for k in net.trainable_order) == 113316
    check_shared_storage(model, learner)


def test_pg_learner_takes_the_references_actor(ref):
    from xuance_amd.learners import PG_Learner
    init, nn = torch.nn.init.orthogonal_, ref.nn
    rep = ref.Basic_MLP((4,), [128], None, init, nn.LeakyReLU, "cpu")
    actor = ref.CategoricalActor(representation=rep, action_space=ref.sp.Discrete(2), actor_hidden_size=[128], normalizer=None,
                                 initializer=init, activation=nn.LeakyReLU, device="cpu")
    model = ref.VanillaPolicyGradient(actor=actor)
    learner = PG_Learner(cfg(n_epochs=1, n_minibatch=1), model, None)
    assert learner.model.head_ld == 2 and learner.model.activation == "leaky_relu"
    check_shared_storage(model, learner)


def _mixing(ref, mixer_cls, use_rnn):
    init, nn = torch.nn.init.orthogonal_, ref.nn
    N, O, S, A = 3, 30, 48, 9
    keys = [f"agent_{i}" for i in range(N)]
    grouping = ref.AgentGrouping.shared(keys)
    group = grouping.group_keys[0]
    if use_rnn:
        obs_rep = ref.Basic_RNN((O,), None, None, init, nn.ReLU, "cpu", fc_hidden_sizes=[64], recurrent_hidden_size=64,
                                N_recurrent_layers=1, dropout=0, rnn="GRU")
    else:
        obs_rep = ref.Basic_MLP((O,), [64], None, init, nn.ReLU, "cpu")
    ident = ref.build_identity_encoder(num_identities=N, mode="none", embedding_dim=None, device="cpu")
    fusion = ref.IdentityFeatureFusion(observation_feature_dim=64, identity_feature_dim=ident.output_dim, mode="concat")
    rep = ref.AgentFeatureEncoder(representation=obs_rep, identity_encoder=ident, fusion=fusion)
    critic = ref.DiscreteActionValueCritic(representation=rep, action_space=ref.sp.Discrete(A), critic_hidden_size=[64],
                                           normalizer=None, initializer=init, activation=nn.ReLU, device="cpu")
    mixer = ref.QMIX_Mixer(S, 32, 32, N, "cpu") if mixer_cls == "QMIX" else ref.VDN_Mixer()
    return grouping, ref.MixingQNetwork(grouping, nn.ModuleDict({group: critic}), mixer, use_rnn=use_rnn, device="cpu")


@pytest.mark.parametrize("use_rnn", [False, True])
def test_qmix_learner_takes_the_references_mixing_q_network(ref, use_rnn):
    from xuance_amd.learners import QMIX_Learner
    grouping, model = _mixing(ref, "QMIX", use_rnn)
    learner = QMIX_Learner(cfg(use_rnn=use_rnn, learning_rate=7e-4), grouping, model, None)    # the reference's call shape
    net = learner.model
    assert (net.n_agents, net.obs_dim, net.n_actions, net.state_dim, net.H, net.HH) == (3, 30, 9, 48, 32, 32)
    assert net.use_rnn == use_rnn and learner.agent_keys == ["agent_0", "agent_1", "agent_2"]
    assert sum(int(np.prod(net.params.shapes[k])) for k in net.trainable_order) == (42218 if use_rnn else 17258)
    check_shared_storage(model, learner)


def test_vdn_learner_takes_a_parameter_free_mixer(ref):
    from xuance_amd.learners import VDN_Learner
    grouping, model = _mixing(ref, "VDN", False)
    learner = VDN_Learner(cfg(), grouping, model, None)
    assert learner.model.mixer == "VDN" and learner.model.n_agents == 3
    check_shared_storage(model, learner)


def test_adoption_refuses_what_the_engine_does_not_run(ref):
    from xuance_amd.adapters import adopt, AdoptError
    with pytest.raises(AdoptError, match="unrecognised policy"):
        adopt(torch.nn.Linear(3, 3), cfg())
    with pytest.raises(AdoptError):
        adopt({}, cfg())
