"""TEST INFRASTRUCTURE ONLY -- the multi-GPU parity oracle of SURVEY.md section 8(e): TWO reference processes with a gloo group and
torch's DistributedDataParallel around the trained modules, on CPU, each rank updating on its OWN batch.  Writes
tests/golden/ddp2_{dqn_mlp,qmix_ff_double,ppo_categorical}.npz: the averaged gradients of every update, the parameters after
every update, every rank's info dict.  The batches are the ones of the single-process fixtures of the same name (rank r takes
batch (r + u) % 2 at update u, which is what tests/test_gpu_dist.py hands its ranks); they are not stored again.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_ddp.py

What is the reference's and what is this script's:
  * models, learners, configs, batches: built by oracle/make_golden.py's generators (golden_dqn / golden_qmix / golden_ppo), i.e. the
    UNMODIFIED reference classes; the generators' `run_learner_updates` hook is where this script takes over.
  * DQN: the reference wraps `representation` and `eval_Q_head` in DistributedDataParallel when `use_distributed_training` is set
    (deep_q_network.py:55-59) -- with `device_ids=[rank]`, which torch refuses for CPU modules.  The same two wraps are made here
    without `device_ids` (the CPU form); everything else is DQN_Learner.update as shipped.
  * QMIX: the reference wraps the ModuleDict `individual_q_networks` and the mixer (value_factorization.py:44-48) and then indexes
    the wrapper (`self.individual_q_networks[group]`, :79) -- `'DistributedDataParallel' object is not subscriptable`: its
    distributed QMIX path raises as shipped.  Here each group's critic and the mixer are wrapped (what the wrapping is for).
  * PPO: the reference does not wrap SharedActorCritic at all (actor_critic.py:8-72: the ranks would train unsynchronised replicas);
    SURVEY 8(e) defines the N-rank PPO update as DDP-equivalent -- mean of the ranks' minibatch gradients, clip_grad_norm_ after the
    average (it runs after backward, ppo_learner.py:62-64), per-rank advantage normalisation -- so representation / actor / critic
    are wrapped here the way DeepQNetwork wraps its parts.
The arithmetic of the average itself is torch's (gloo all-reduce of each bucket, divided by the world size)."""
import os
import subprocess
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
WORLD, N_UPDATES = 2, 2
KINDS = ("dqn_mlp", "qmix_ff_double", "ppo_categorical")


def _strip(name):
    return name.replace(".module.", ".")


def rank_main(rank, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    import make_golden as mg                                        # installs the shim, imports the reference
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    save = np.savez_compressed
    captured = {}
    mg.np.savez_compressed = lambda path, **out: captured.update(path=path, out=out)     # (the generators end with a save: keep theirs untouched)
    mg.float64_twin = lambda *a, **k: {}

    def runner(kind):
        def run(learner, model, cb, batches, call):
            committed = np.load(os.path.join(mg.OUT, kind + ".npz"))
            for u in range(2):                                      # same seeds -> the committed fixture's batches
                for k, v in batches[u].items():
                    assert np.array_equal(np.asarray(v), committed[f"u{u}/batch/{k}"]), (kind, u, k)
            for n, v in mg.sd_np(model).items():
                assert np.array_equal(v, committed[f"init/{n}"]), (kind, n)
            if kind == "dqn_mlp":                                   # deep_q_network.py:55-59 (CPU form: no device_ids)
                model.representation = DDP(model.representation)
                model.eval_Q_head = DDP(model.eval_Q_head)
                do = lambda b: learner.update(batch_size=len(b["obs"]), **b)
            elif kind == "qmix_ff_double":                          # value_factorization.py:44-48, on the members (see the header)
                for g in list(model.individual_q_networks.keys()):
                    model.individual_q_networks[g] = DDP(model.individual_q_networks[g])
                model.eval_Qtot = DDP(model.eval_Qtot)
                keys = model.agent_keys

                def do(b):
                    s = {k: {a: b[k][:, i] for i, a in enumerate(keys)}
                         for k in ("obs", "obs_next", "actions", "rewards", "terminals", "agent_mask", "avail_actions", "avail_actions_next")}
                    s.update(state=b["state"], state_next=b["state_next"], batch_size=len(b["state"]))
                    return learner.update(s)
            else:                                                   # DDP added around the actor-critic's parts
                model.representation = DDP(model.representation)
                model.actor = DDP(model.actor)
                model.critic = DDP(model.critic)
                do = lambda b: learner.update(obs=b["obs"], actions=b["actions"], returns=b["returns"], values=b["values"],
                                              advantages=b["advantages"], aux_batch={"old_logp": b["old_logp"]}, batch_size=len(b["obs"]))
            out = {}
            for u in range(N_UPDATES):
                info = do(batches[(rank + u) % 2])
                out.update({f"u{u}/info_rank{rank}/{k}": np.float64(v) for k, v in info.items()
                            if np.isscalar(v) or isinstance(v, (float, int, torch.Tensor))})
                out.update({f"u{u}/grad/{_strip(n)}": p.grad.detach().numpy().copy() for n, p in model.named_parameters() if p.grad is not None})
                out.update({f"u{u}/param/{_strip(k)}": v for k, v in mg.sd_np(model).items()})
            return out
        return run

    for kind in KINDS:
        mg.run_learner_updates = runner(kind)
        captured.clear()
        if kind == "dqn_mlp":
            mg.golden_dqn("mlp")
        elif kind == "qmix_ff_double":
            mg.golden_qmix(True)
        else:
            mg.golden_ppo("categorical")
        out = captured["out"]
        gathered = [None] * WORLD
        dist.all_gather_object(gathered, {k: v for k, v in out.items() if "/info_rank" in k or "/grad/" in k or "/param/" in k})
        if rank == 0:
            mine, other = gathered
            for k, v in mine.items():                               # replicas: same averaged gradients, same parameters
                if "/grad/" in k or "/param/" in k:
                    assert np.array_equal(v, other[k]), (kind, k)
            out.update({k: v for k, v in other.items() if "/info_rank" in k})
            loss = next(k.split("/")[-1] for k in sorted(out) if k.startswith("u0/info_rank0/") and "loss" in k.lower())
            assert out["u0/info_rank0/" + loss] != out["u0/info_rank1/" + loss], "the ranks must have seen different batches"
            out["n_updates"], out["world"] = np.int64(N_UPDATES), np.int64(WORLD)
            save(os.path.join(mg.OUT, f"ddp2_{kind}.npz"), **out)
            print(f"ddp2_{kind}.npz", len(out), "arrays", flush=True)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 3:
        rank_main(int(sys.argv[1]), int(sys.argv[2]))
        sys.exit(0)
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(port)], env=env) for r in range(WORLD)]
    rc = [p.wait() for p in procs]
    assert rc == [0] * WORLD, rc
