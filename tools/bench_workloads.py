"""MEASUREMENT INFRASTRUCTURE -- the workloads `bench.py --workload` can time on N ranks (one process per GPU), and the ways the
ranks can average their gradients.  Workloads = the BASELINE.json configurations the north-star sizes for a node:

  c2      PPO-Clip CartPole-v1, 256 envs/GPU x horizon 256, 8 x 8 minibatches of 8 192 (configs[1]; the headline)
  c4      PPO, HalfCheetah shapes (Gaussian 17-256-256-6 + critic), 128 envs/GPU x 256 (configs[3]: 1 024 envs over 8 GPUs)
  qmix3m  QMIX, SMAC-3m shape, 64 envs/GPU, feed-forward agents, batch 32, 8 updates per vector step (configs[4]: 512 envs over
          8 GPUs; the reference DDP-wraps exactly these modules, value_factorization.py:44-48)
  qmix3m_gru  the same with the recurrent agents of configs/qmix/sc2/3m.yaml

Gradient paths (xuance_amd/dist.py): "exchange" = averaged inside the optimiser launch through IPC-mapped
peer buffers, "captured" = the process group's all-reduce captured in the update graph, "cut" = update graphs cut at the
collectives (off-policy learners: eager launches around the all-reduce).  `usable_paths` asks the start-up self-tests (collective
calls: every rank must make them), `Runner` builds one agent for one path."""
import os
from argparse import Namespace

import torch

WORKLOADS = ("c2", "c4", "qmix3m", "qmix3m_gru")


def usable_paths(workload, device):
    """The gradient paths this job can use for `workload`, fastest-expected first.  Collective (self-tests)."""
    from xuance_amd import dist as xd
    out = []
    import torch.distributed as dist
    shared = torch.cuda.device_count() < dist.get_world_size()     # ranks time-share a GPU (test boxes only)
    # (the 558 optimiser blocks of the c4 network spin on their peers: ranks sharing ONE GPU cannot all be resident)
    if xd.exchange_usable(device) and not (shared and workload == "c4"):
        out.append("exchange")
    if workload in ("c2", "c4") and xd.collective_capturable(device):
        out.append("captured")
    out.append("cut")
    return out


def _path_cfg(path):
    return {"auto": {}, "exchange": dict(dist_gradient_exchange=True),
            "captured": dict(dist_gradient_exchange=False, dist_graph_collective=True),
            "cut": dict(dist_gradient_exchange=False, dist_graph_collective=False)}[path]


def c2_config(n_envs, horizon, world, rank):
    return Namespace(agent="PPO", env_id="CartPole-v1", representation="Basic_MLP", representation_hidden_size=[128],
                     actor_hidden_size=[128], critic_hidden_size=[128], activation="leaky_relu", seed=1 + rank,
                     parallels=n_envs, running_steps=10 ** 9, horizon_size=horizon, n_epochs=8, n_minibatch=8,
                     learning_rate=4e-4, vf_coef=0.25, ent_coef=0.01, clip_range=0.2, gamma=0.98, use_gae=True,
                     gae_lambda=0.95, use_advnorm=True, use_grad_clip=True, grad_clip_norm=0.5, use_obsnorm=True,
                     use_rewnorm=True, obsnorm_range=5, rewnorm_range=5, distributed_training=world > 1, device="cuda",
                     model_dir="/tmp/xrl_bench_models", use_hip_graph=True)


def c4_config(n, T, world, rank):
    return Namespace(agent="PPO", representation="Basic_Identical", representation_hidden_size=[], actor_hidden_size=[256, 256],
                     critic_hidden_size=[256, 256], activation="leaky_relu", activation_action="tanh", seed=1 + rank, parallels=n,
                     running_steps=10 ** 9, horizon_size=T, n_epochs=16, n_minibatch=8, learning_rate=4e-4, vf_coef=0.25,
                     ent_coef=0.0, clip_range=0.2, gamma=0.99, use_gae=True, gae_lambda=0.95, use_advnorm=True, use_grad_clip=True,
                     grad_clip_norm=0.5, use_obsnorm=True, use_rewnorm=True, obsnorm_range=5, rewnorm_range=5,
                     distributed_training=world > 1, device="cuda", model_dir="/tmp/xrl_bench_models", use_hip_graph=True)


def qmix_config(n, rnn, world, rank):
    c = dict(q_hidden_size=[64], hidden_dim_mixing_net=32, hidden_dim_hyper_net=32, activation="relu", seed=1 + rank, parallels=n,
             running_steps=10 ** 7, batch_size=32, learning_rate=7e-4, gamma=0.99, double_q=True, start_greedy=1.0,
             end_greedy=0.05, decay_step_greedy=50000, sync_frequency=200, training_frequency=1, n_epochs=8,
             use_grad_clip=False, grad_clip_norm=10.0, use_actions_mask=True, use_parameter_sharing=True, use_rnn=rnn,
             distributed_training=world > 1, device="cuda", model_dir="/tmp/xrl_bench_models")
    if rnn:   # configs/qmix/sc2/3m.yaml defaults; rnn_backprop_agents False = the reference's behaviour (agents detached)
        c.update(fc_hidden_sizes=[64], recurrent_hidden_size=64, buffer_size=5000, start_training=1000, rnn_backprop_agents=False,
                 episode_length=60)
    else:
        c.update(representation_hidden_size=[64], buffer_size=n * 78, start_training=640)
    return Namespace(**c)


class Runner:
    """One agent of one workload on this rank, one gradient path.  step() = one pass of the hot path over one batch."""

    def __init__(self, workload, world, rank, path="auto", n_envs=None, horizon=256, extra=None):
        from xuance_amd.agents import PPO_Agent, QMIX_Agents
        from xuance_amd.envs import DeviceCartPoleVecEnv, SyntheticMujocoVecEnv, SyntheticSMACVecEnv
        self.workload, self.world, self.rank, self.path = workload, world, rank, path
        torch.manual_seed(1)                               # same initial parameters on every rank (and rank 0's are broadcast)
        if workload == "c2":
            self.n = n_envs or 256
            cfg = c2_config(self.n, horizon, world, rank)
            env = DeviceCartPoleVecEnv(self.n, seed=1 + rank)
            self.steps_per_pass = self.n * horizon
            self.metric = "env-steps/sec (rollout+update), PPO-Clip CartPole-v1"
            self.describe = ("PPO-Clip CartPole-v1, %d envs/GPU x horizon %d, 8 epochs x 8 minibatches of %d, net 4-128-{128-2,128-1} "
                             "(BASELINE.json configs[1])" % (self.n, horizon, self.n * horizon // 8))
            self.env_name = "device-resident CartPole-v1 (xrl_cartpole_step)"
        elif workload == "c4":
            self.n = n_envs or 128
            cfg = c4_config(self.n, horizon, world, rank)
            env = SyntheticMujocoVecEnv(self.n, seed=4 + rank)
            self.steps_per_pass = self.n * horizon
            self.metric = "env-steps/sec (rollout+update), PPO-Clip HalfCheetah shapes"
            self.describe = ("PPO, HalfCheetah shapes (obs 17, Box(6), Gaussian 17-256-256-6 + critic 17-256-256-1), %d envs/GPU x horizon %d, "
                             "16 epochs x 8 minibatches of %d (BASELINE.json configs[3]: 1 024 envs over 8 GPUs)" % (self.n, horizon, self.n * horizon // 8))
            self.env_name = "MuJoCo-shaped synthetic provider on the device (xrl_synth_control_step; no simulator in the image)"
        else:
            rnn = workload == "qmix3m_gru"
            self.n = n_envs or 64
            cfg = qmix_config(self.n, rnn, world, rank)
            env = SyntheticSMACVecEnv(self.n, seed=3 + rank)
            self.vec_steps = 60 if rnn else 16
            self.steps_per_pass = None                      # counted from agent.current_step (episodes end at their own pace)
            self.metric = "env-steps/sec (acting+replay+update), QMIX SMAC-3m shape"
            self.describe = ("QMIX SMAC-3m shape, %d envs/GPU x 3 agents, obs 30 / state 48 / 9 masked actions, %s, batch 32, 8 updates per %s "
                             "(BASELINE.json configs[4]: 512 envs over 8 GPUs); one pass = agent.train(%d)"
                             % (self.n, "recurrent agents (3m.yaml)" if rnn else "feed-forward agents",
                                "%d episodes" % self.n if rnn else "vector step", self.vec_steps))
            self.env_name = "SMAC-3m-shaped synthetic provider on the device (xrl_synth_marl_step; no simulator in the image)"
        for k, v in dict(_path_cfg(path), **(extra or {})).items():
            setattr(cfg, k, v)
        if world > torch.cuda.device_count() and workload == "c4":
            # ranks time-sharing ONE GPU (test boxes): the 279 blocks of this network's one-launch optimiser step spin on a
            # barrier and need every block resident, which a second process on the device can prevent -- two launches instead
            cfg.use_fused_optimizer = False
            cfg.use_wide_rollout = False                    # (likewise the whole-rollout launch: its workgroups wait for each other)
        self.ppo = workload in ("c2", "c4")
        self.agent = (PPO_Agent if self.ppo else QMIX_Agents)(cfg, env)
        if world > 1:
            from xuance_amd.dist import broadcast_
            broadcast_(self.agent.model.params.flat, 0)
            if getattr(self.agent.model, "target_flat", None) is not None:
                broadcast_(self.agent.model.target_flat, 0)
        self.info = {}
        self._s0 = 0

    def step(self):
        if self.ppo:
            self.agent.rollout()
            self.info = self.agent.update()
        else:
            self.info = self.agent.train(self.vec_steps)

    def mark(self):
        self._s0 = getattr(self.agent, "current_step", 0)

    def env_steps_since_mark(self, passes):
        """Env steps THIS rank made in `passes` passes since mark()."""
        if self.steps_per_pass is not None:
            return self.steps_per_pass * passes
        return int(self.agent.current_step - self._s0)

    def gradient_average(self):
        lr = self.agent.learner
        if self.world == 1:
            return None
        if getattr(lr, "_xc", None) is not None:
            return "inside the optimiser launch through IPC-mapped exchange buffers (xrl_reduce_adam_exchange; no collective call on the data path)"
        if self.ppo:
            return ("process-group all-reduce captured in the update graph" if getattr(self.agent, "_whole_phase_graph", False)
                    else "process-group all-reduce between update graphs cut at the collectives")
        return "process-group all-reduce between eager launches of the update"

    def rollout_mode(self):
        """How the timed rollouts ran (c2: csrc/rollout_actor.hip reports its placement per launch)."""
        a = self.agent
        if self.workload == "c2":
            if getattr(a, "persist_status", None) is None:
                return "one launch per vector step (%s)" % ("xrl_rollout_cartpole_run, n_steps = 1" if a._actor_rollout() is not None
                                                             else "xrl_rollout_step_cartpole")
            st = [int(x) for x in getattr(a.learner, "last_status", None) or a.persist_status.tolist()]
            st += [0] * (4 - len(st))
            return ("whole-rollout launch (xrl_rollout_cartpole_run: actor-only step chain) + batched values launch, status %s: %s" %
                    (st, "workgroups on ONE XCD, plain-store messages through its L2" if st[3] == 0 else
                     "%d launch(es) found their workgroups on several XCDs and exchanged through device-scope stores" % st[3]))
        if self.workload == "c4":
            if getattr(a, "_wr", None) is not None:
                return "whole-rollout launch (xrl_rollout_wide_run: actor-only step chain, dynamics inside) + batched values passes"
            return "two launches per vector step (xrl_wide_act_step incl. statistics + bookkeeping; provider)" if a._wide_acting() is not None \
                else "layered launches per vector step"
        return "one acting launch + provider + store per vector step; updates as one graph per phase" if getattr(a.learner, "_buf_graph", None) is not None \
            else "one acting launch + provider + store per vector step; eager update launches"

    def close(self):
        xc = getattr(self.agent.learner, "_xc", None)
        if xc is not None:
            try:
                xc.close()
            except Exception:                               # noqa: BLE001
                pass
        self.agent = None
        torch.cuda.empty_cache()


def fence(world):
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed(runner, steps, warmup, world):
    """`warmup` untimed passes, then exactly `steps` timed ones bracketed by barrier + synchronize on both sides; the MAX over
    ranks of the elapsed time, the SUM over ranks of the env steps."""
    import time
    import torch.distributed as dist
    import gc
    for _ in range(warmup):
        runner.step()
    # Objects of earlier measurements (the runners of measure_paths, their captured graphs) are destroyed NOW, and the collector
    # stays off inside the timed region: destroying a graph's executable stalls the host for tens of milliseconds.  Nothing of the
    # K steps is skipped by this -- the steps allocate no cyclic garbage worth collecting.
    gc.collect()
    gc.disable()
    try:
        fence(world)
        runner.mark()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step()
        fence(world)
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    n = runner.env_steps_since_mark(steps)
    if world > 1:
        t = torch.tensor([elapsed, -float(n)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        s = torch.tensor([float(n)], dtype=torch.float64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        elapsed, n = float(t[0].item()), int(s.item())
    return elapsed, n


def measure_paths(workload, world, rank, device, n_envs=None, horizon=256, passes=2, only=None):
    """Every usable gradient path timed on the real workload (1 warm-up pass + `passes` timed ones, max over ranks), so that the
    one adopted is the fastest MEASURED here, not the first whose self-test passed.  -> (name of the fastest, {name: ms per
    pass or an error string})."""
    import torch.distributed as dist
    paths = usable_paths(workload, device)
    if only:
        paths = [p for p in paths if p in only] or paths[-1:]
    res = {}
    for p in paths:
        ok, ms = 1.0, None
        try:
            r = Runner(workload, world, rank, p, n_envs, horizon)
            el, _ = timed(r, passes, 1, world)
            ms = el / passes * 1e3
            r.close()
        except Exception as ex:                              # noqa: BLE001
            ok, ms = 0.0, repr(ex)[:200]
        flag = torch.tensor([ok], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res[p] = round(ms, 4) if (flag.item() == 1.0 and not isinstance(ms, str)) else ("failed: %s" % (ms if isinstance(ms, str) else "on another rank"))
    good = {k: v for k, v in res.items() if not isinstance(v, str)}
    best = min(good, key=good.get) if good else "cut"
    return best, res


def rccl_world(world):
    """A one-element all-reduce over the process group: must come back as the number of ranks."""
    import torch.distributed as dist
    if world == 1:
        return 1
    x = torch.ones(1, device="cuda")
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return int(x.item())


def xgmi_note():
    return os.environ.get("XRL_DIST_BACKEND", "nccl")
