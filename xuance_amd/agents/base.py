"""What the reference's runner touches on an agent besides ``train`` (xuance/engine/run_drl.py:101-203):
``distributed_training``, ``model_dir_load`` / ``model_dir_save``, ``meta_data``, ``save_model(model_name, model_path)``,
``load_model(path, model)``, ``test(test_episodes, test_envs, close_envs)``, ``current_step``, ``config``, ``finish()`` --
the surface of xuance/torch/agents/base/agent.py:60-233, 344-359 and of the ``test`` methods of core/on_policy.py:303-400,
core/off_policy.py:272-350, core/off_policy_marl.py:596-640, shared by the HIP agents through this mixin.

The evaluation environments of ``test`` are whatever the runner built with ``make_envs``: HOST vector envs with the
reference contract (``reset() -> (obs, infos)``, ``step(actions) -> (obs, rewards, terminals, truncations, infos)`` with
``infos[i]["reset_obs"]`` / ``["episode_score"]`` at episode ends; dummy_vec_env.py:46-76).  Observations go up, actions
come down once per step; the policy forward / action selection runs in the same kernels as the training loop."""
import os
import time
from collections import namedtuple

import numpy as np
import torch

from .. import __name__ as _pkg

ActionOutput = namedtuple("ActionOutput", ["env_actions", "values", "distributions", "log_probs"])     # outputs.py: ActionOutput


def _get(cfg, name, default=None):
    return getattr(cfg, name, default)


class AgentSurface:
    """Mixin: call ``_init_surface()`` from the constructor once ``config`` / ``envs`` / ``device`` are set."""

    def _init_surface(self):
        c = self.config
        self.distributed_training = bool(_get(c, "distributed_training", False))
        self.world_size = int(os.environ.get("WORLD_SIZE", 1)) if self.distributed_training else 1
        self.rank = int(os.environ.get("RANK", 0)) if self.distributed_training else 0
        self.render, self.fps = _get(c, "render", False), _get(c, "fps", 50)
        self.atari = _get(c, "env_name", "") == "Atari"
        self.train_envs = self.envs
        if self.envs is not None and getattr(self.envs, "max_episode_steps", None) is not None:
            self.episode_length = c.episode_length = self.envs.max_episode_steps       # agent.py:108
        stamp = time.strftime("%Y_%m%d_%H%M%S")                         # get_time_string (common_tools.py)
        model_dir = _get(c, "model_dir", "models")
        self.model_dir_load = model_dir                                 # agent.py:149-150
        self.model_dir_save = os.path.join(os.getcwd(), model_dir, f"seed_{_get(c, 'seed', 1)}_{stamp}")
        self.log_dir = os.path.join(os.getcwd(), _get(c, "log_dir", "logs"), f"seed_{_get(c, 'seed', 1)}_{stamp}")
        self.meta_data = dict(algo=_get(c, "agent", type(self).__name__), env=_get(c, "env_name", None),
                              env_id=_get(c, "env_id", None), dl_toolbox="torch (xuance_amd: HIP gfx950)", device=str(self.device),
                              seed=_get(c, "seed", 1), xuance_version=_get(c, "xuance_version", _pkg))       # agent.py:195-197
        self.logged = []                                                # log_infos sink: (step, dict) pairs

    # -- the loop hooks of xuance/common/callback.py:31-58 --------------------------------------------------------------------
    def _cb(self, hook, *args, **kwargs):
        """Call `hook` of the user's callback if it has one (None / the learners' null object: nothing happens).  The device
        loops call: on_train_step / on_train_step_end once per VECTOR step where the loop is a host loop anyway (DQN, feed-forward
        QMIX; PPO with `per_step_callbacks: True`, which runs the rollout as per-step launches), otherwise once per ROLLOUT / per
        run_episodes call with `steps=<vector steps covered>` in the kwargs; on_train_epochs_end after every update phase.  Tensors
        handed over are the device tensors of the buffers (no copies, no host sync unless the callback reads them)."""
        fn = getattr(self.callback, hook, None) if getattr(self, "callback", None) is not None else None
        if fn is not None:
            return fn(*args, **kwargs)

    def _has_cb(self, hook):
        """Does the user's callback OVERRIDE `hook` (a BaseCallback subclass that leaves it alone does not count)?"""
        cb = getattr(self, "callback", None)
        fn = getattr(type(cb), hook, None) if cb is not None else None
        if fn is None:
            return False
        for base in type(cb).__mro__[1:]:
            if base.__name__ == "BaseCallback" and getattr(base, hook, None) is fn:
                return False
        return True

    # -- logging hooks the loops call (agent.py:235-262); the reference writes tensorboard / wandb, out of scope here
    def log_infos(self, info, x_index):
        self.logged.append((int(x_index), dict(info)))

    # -- checkpoints (agent.py:199-233) -----------------------------------------------------------------------------
    def _obs_stats_tensors(self):
        """(mean, var, count) device tensors of the observation statistics, or None when the agent keeps none."""
        if hasattr(self, "obs_mean"):
            return self.obs_mean, self.obs_var, self.obs_count
        return None

    def save_model(self, model_name, model_path=None):
        if self.distributed_training and self.rank > 0:
            return
        model_path = self.model_dir_save if model_path is None else model_path
        os.makedirs(model_path, exist_ok=True)
        self.learner.save_model(os.path.join(model_path, model_name))
        st = self._obs_stats_tensors() if _get(self.config, "use_obsnorm", False) else None
        if st is not None:
            from ..spaces import space2shape
            mean, var, count = st
            shape = space2shape(self.observation_space)
            np.save(os.path.join(model_path, "obs_rms.npy"),
                    {"count": float(count.item()), "mean": mean.cpu().numpy().astype(np.float32).reshape(shape),
                     "var": var.cpu().numpy().astype(np.float32).reshape(shape)})

    def load_model(self, path, model=None):
        loaded_dir = self.learner.load_model(path, model)               # the directory the file was found in (:95-157)
        st = self._obs_stats_tensors() if _get(self.config, "use_obsnorm", False) else None
        if st is not None:
            f = os.path.join(loaded_dir, "obs_rms.npy")
            if not os.path.exists(f):
                raise RuntimeError(f"Failed to load observation status file 'obs_rms.npy' from {f}!")
            rec = np.load(f, allow_pickle=True).item()
            mean, var, count = st
            mean.copy_(torch.as_tensor(np.asarray(rec["mean"], np.float32).reshape(-1)))
            var.copy_(torch.as_tensor(np.asarray(rec["var"], np.float32).reshape(-1)))
            count.fill_(float(rec["count"]))
        self._after_load()
        return loaded_dir

    def _after_load(self):
        pass

    def finish(self):
        if self.envs is not None:
            self.envs.close()

    # -- evaluation: the loop of on_policy.py:344-385 / off_policy.py:300-340 over HOST vector envs ---------------------
    def _test_actions(self, obs, deterministic):
        """obs: [num_envs, ...] NumPy (raw) -> actions NumPy.  Subclasses: normalise + forward + select on the device."""
        raise NotImplementedError

    def test(self, test_episodes, deterministic_policy=True, test_envs=None, close_envs=True):
        if test_envs is None:
            raise ValueError("`test_envs` must be provided for evaluation.")
        num_envs = test_envs.num_envs
        current_episode, current_step, scores, best_score = 0, 0, [], -np.inf
        obs, infos = test_envs.reset()
        while current_episode < test_episodes:
            acts = self._test_actions(np.asarray(obs), deterministic_policy)
            next_obs, rewards, terminals, truncations, infos = test_envs.step(acts)
            obs = np.array(next_obs, copy=True)
            for i in range(num_envs):
                if terminals[i] or truncations[i]:
                    if self.atari and (not truncations[i]):
                        continue                                        # life loss (on_policy.py:366-367)
                    obs[i] = infos[i]["reset_obs"]
                    scores.append(infos[i]["episode_score"])
                    current_episode += 1
                    best_score = max(best_score, infos[i]["episode_score"])
            current_step += num_envs
        self.log_infos({"Test-Episode-Rewards/Mean-Score": float(np.mean(scores)),
                        "Test-Episode-Rewards/Std-Score": float(np.std(scores))}, self.current_step)
        if close_envs:
            test_envs.close()
        return scores
