"""TEST INFRASTRUCTURE / DOCUMENTATION ONLY -- times the UNMODIFIED reference's learner updates on this container's CPU
(through oracle/ref_shim.py) at the BASELINE shapes, as context for the device numbers in DESIGN.md.  Not used by any
test or by bench.py (the reference does not exist on the GPU box).   PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_cpu.py
"""
import os, sys, time, json
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import importlib.util
spec = importlib.util.spec_from_file_location("mg", os.path.join(HERE, "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
import numpy as np
import torch
from torch import nn

torch.set_num_threads(os.cpu_count())
sp, rng = mg.sp, np.random.default_rng(0)
init = torch.nn.init.orthogonal_


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def qmix(rnn):
    from xuance.torch.rl_models.critics.base_critics import DiscreteActionValueCritic
    from xuance.torch.rl_models.representations.agent_feature import AgentFeatureEncoder
    from xuance.torch.rl_models.representations import Basic_RNN
    from xuance.torch.rl_models.modules.identity_encoder import build_identity_encoder, IdentityFeatureFusion
    N, O, S, A, B, T = 3, 30, 48, 9, 32, 60
    keys = [f"agent_{i}" for i in range(N)]
    grouping = mg.AgentGrouping.shared(keys)
    group = grouping.group_keys[0]
    if rnn:
        rep0 = Basic_RNN((O,), None, None, init, nn.ReLU, "cpu", fc_hidden_sizes=[64], recurrent_hidden_size=64,
                         N_recurrent_layers=1, dropout=0, rnn="GRU")
    else:
        rep0 = mg.Basic_MLP((O,), [64], None, init, nn.ReLU, "cpu")
    ident = build_identity_encoder(num_identities=N, mode="none", embedding_dim=None, device="cpu")
    fusion = IdentityFeatureFusion(observation_feature_dim=64, identity_feature_dim=ident.output_dim, mode="concat")
    critic = DiscreteActionValueCritic(representation=AgentFeatureEncoder(representation=rep0, identity_encoder=ident, fusion=fusion),
                                       action_space=sp.Discrete(A), critic_hidden_size=[64], normalizer=None, initializer=init,
                                       activation=nn.ReLU, device="cpu")
    model = mg.MixingQNetwork(grouping, nn.ModuleDict({group: critic}), mg.QMIX_Mixer(S, 32, 32, N, "cpu"), use_rnn=rnn, device="cpu")
    cfg = mg.base_config(learning_rate=7e-4, gamma=0.99, sync_frequency=200, start_training=0, training_frequency=1,
                         use_parameter_sharing=True, double_q=True, use_actions_mask=not rnn, use_rnn=rnn, n_epochs=8,
                         use_grad_clip=False, episode_length=T, parallels=64)
    learner = mg.QMIX_Learner(cfg, grouping, model, mg.Capture())
    if rnn:
        b = dict(obs=rng.standard_normal((B, N, T + 1, O)).astype(np.float32), actions=rng.integers(0, A, (B, N, T)).astype(np.float32),
                 rewards=rng.standard_normal((B, N, T)).astype(np.float32), terminals=np.zeros((B, N, T), bool),
                 agent_mask=np.ones((B, N, T), bool))
        sample = {k: {a: b[k][:, i] for i, a in enumerate(keys)} for k in b}
        sample.update(state=rng.standard_normal((B, T + 1, S)).astype(np.float32), filled=np.ones((B, T), bool), batch_size=B,
                      sequence_length=T)
    else:
        b = dict(obs=rng.standard_normal((B, N, O)).astype(np.float32), obs_next=rng.standard_normal((B, N, O)).astype(np.float32),
                 actions=np.zeros((B, N), np.float32), rewards=rng.standard_normal((B, N)).astype(np.float32),
                 terminals=np.zeros((B, N), bool), agent_mask=np.ones((B, N), bool), avail_actions=np.ones((B, N, A), bool),
                 avail_actions_next=np.ones((B, N, A), bool))
        sample = {k: {a: b[k][:, i] for i, a in enumerate(keys)} for k in b}
        sample.update(state=rng.standard_normal((B, S)).astype(np.float32), state_next=rng.standard_normal((B, S)).astype(np.float32),
                      batch_size=B)
    return timed(lambda: learner.update(sample), 20)


def dqn_cnn():
    rep = mg.Basic_CNN((84, 84, 4), [8, 4, 3], [4, 2, 1], [32, 64, 64], None, init, nn.ReLU, "cpu")
    model = mg.DeepQNetwork(rep, [512], sp.Discrete(4), None, init, nn.ReLU, "cpu")
    cfg = mg.base_config(learning_rate=1e-4, gamma=0.99, sync_frequency=500, start_training=0, training_frequency=1,
                         use_grad_clip=False)
    learner = mg.DQN_Learner(cfg, model, mg.Capture())
    bs = 32
    b = dict(obs=rng.integers(0, 256, (bs, 84, 84, 4)).astype(np.uint8), obs_next=rng.integers(0, 256, (bs, 84, 84, 4)).astype(np.uint8),
             actions=rng.integers(0, 4, bs).astype(np.float32), rewards=rng.standard_normal(bs).astype(np.float32),
             terminals=np.zeros(bs, np.float32))
    return timed(lambda: learner.update(batch_size=bs, **b), 10)


def ppo(bs):
    rep = mg.Basic_MLP((4,), [128], None, init, nn.LeakyReLU, "cpu")
    model = mg.SharedActorCritic(rep, mg.CategoricalActorHead(128, [128], 2, None, init, nn.LeakyReLU, "cpu"),
                                 mg.ValueHead(128, [128], None, init, nn.LeakyReLU, "cpu"))
    cfg = mg.base_config(horizon_size=256, n_epochs=8, n_minibatch=8, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, parallels=256)
    learner = mg.PPO_Learner(cfg, model, mg.Capture())
    b = dict(obs=rng.standard_normal((bs, 4)).astype(np.float32), actions=rng.integers(0, 2, bs).astype(np.float32),
             returns=rng.standard_normal(bs).astype(np.float32), advantages=rng.standard_normal(bs).astype(np.float32),
             values=rng.standard_normal(bs).astype(np.float32), aux_batch={"old_logp": np.full(bs, -0.69, np.float32)}, batch_size=bs)
    return timed(lambda: learner.update(**b), 20)


if __name__ == "__main__":
    out = {"threads": os.cpu_count(),
           "ppo_update_bs8192_ms": round(ppo(8192), 3),
           "qmix_ff_update_b32_ms": round(qmix(False), 3),
           "qmix_rnn_update_b32x60_ms": round(qmix(True), 3),
           "dqn_cnn_update_b32_ms": round(dqn_cnn(), 3)}
    print(json.dumps(out))
