"""IMAmpAgent: the registered algo name ``im_amp`` (phc/run_hydra.py:266) of every imitation config.

Mirrors phc/learning/im_amp.py:39-132 on top of AMPAgent:
  get_action            :44-76   deterministic / sampled action for evaluation loops
  env_eval_step         :78-100
  restore               :102-118 checkpoint + the newest termination history (failed_*.pkl) pushed into the motion library's
                                 sampling probabilities
  update_training_data  :127-132 PMCP: hard / soft re-weighting of the motion sampling from the keys that failed evaluation
  eval                  :136-363 evaluation over the motion library.  The reference computes MPJPE-style metrics with smpl_sim's
                                 compute_metrics_lite (a third-party dependency that is absent); here the evaluation loop reports
                                 success rate and mean global / root-relative body-position errors from the same rigid-body tensors.
"""
import glob
import os
import os.path as osp
import pickle

import torch

from .amp_agent import AMPAgent


class IMAmpAgent(AMPAgent):
    def __init__(self, base_name, config):
        super().__init__(base_name, config)
        self.network_path = config.get("network_path", osp.join(config.get("train_dir", "output/pulse_amd"), "nn"))
        self.has_batch_dimension = True
        self.is_tensor_obses = True

    # ------------------------------------------------------------------ im_amp.py:44-76
    def get_action(self, obs_dict, is_determenistic=False):
        obs = obs_dict["obs"] if isinstance(obs_dict, dict) else obs_dict
        n = obs.shape[0]
        net = self.model
        ws = net.workspace(n, train=False)
        net.eval()
        self.running_mean_std.eval()
        self._preproc_obs(self._obs_store(obs), ws, n)
        net.forward(ws, n)
        mu = ws["mu"]
        if is_determenistic:
            current_action = mu.clone()
        else:
            noise = torch.randn(n, self.actions_num, device=self.ppo_device, generator=self.noise_generator)
            current_action = mu + torch.exp(net.sigma) * noise
        if self.clip_actions:
            d, m = (self.actions_high - self.actions_low) / 2.0, (self.actions_high + self.actions_low) / 2.0
            return torch.clamp(current_action, -1.0, 1.0) * d + m                 # rescale_actions
        return current_action

    def env_eval_step(self, env, actions):
        obs, rewards, dones, infos = env.step(actions)
        return obs, rewards.to(self.ppo_device), dones.to(self.ppo_device), infos

    # ------------------------------------------------------------------ im_amp.py:102-132
    def restore(self, fn):
        super().restore(fn)
        fails = glob.glob(osp.join(self.network_path, "failed_*"))
        if fails:
            newest = sorted(fails, key=lambda x: int(x.split("_")[-1].split(".")[0]))[-1]
            with open(newest, "rb") as f:
                history = pickle.load(f)["termination_history"]
            self.vec_env.env.task._motion_lib.update_sampling_prob(history)

    def update_training_data(self, failed_keys):
        task = self.vec_env.env.task
        lib = task._motion_lib
        # (no hasattr guards: a motion source without the PMCP hooks fails loudly instead of silently training on the old weights)
        if task.auto_pmcp:
            lib.update_hard_sampling_weight(failed_keys)
        elif task.auto_pmcp_soft:
            lib.update_soft_sampling_weight(failed_keys)
        os.makedirs(self.network_path, exist_ok=True)
        with open(osp.join(self.network_path, f"failed_{self.epoch_num:010d}.pkl"), "wb") as f:
            pickle.dump({"failed_keys": failed_keys, "termination_history": lib._termination_history}, f)

    # ------------------------------------------------------------------ im_amp.py:136-363 (metrics: see module docstring)
    def eval(self, max_steps=None):
        """Every env plays its motion once with deterministic actions; returns success rate and position errors."""
        task = self.vec_env.env.task
        n = task.num_envs
        max_steps = int(max_steps or task.max_episode_length)
        self.set_eval()
        obs = self.vec_env.reset()
        done_once = torch.zeros(n, dtype=torch.bool, device=self.ppo_device)
        failed = torch.zeros(n, dtype=torch.bool, device=self.ppo_device)
        err_g = torch.zeros(n, device=self.ppo_device)
        err_l = torch.zeros(n, device=self.ppo_device)
        steps = torch.zeros(n, device=self.ppo_device)
        for _ in range(max_steps):
            act = self.get_action({"obs": obs}, is_determenistic=True)
            obs, _, dones, infos = self.env_eval_step(self.vec_env, act)
            live = ~done_once
            ref = task._track["rb_records"][..., 0:3] if getattr(task, "_use_motion_lib", False) else task.sim.rigid_body_state[..., 0:3]
            cur = task.sim.rigid_body_state[..., 0:3]
            err_g += live * (cur - ref).norm(dim=-1).mean(-1)
            err_l += live * ((cur - cur[:, :1]) - (ref - ref[:, :1])).norm(dim=-1).mean(-1)
            steps += live
            failed |= live & (infos["terminate"] > 0)
            done_once |= dones > 0
            obs = self.vec_env.reset_masked(dones > 0)
        steps = steps.clamp(min=1)
        return {"success_rate": float(1.0 - failed.float().mean()), "mpjpe_g": float((err_g / steps).mean() * 1000.0),
                "mpjpe_l": float((err_l / steps).mean() * 1000.0), "failed_keys": torch.nonzero(failed).flatten().tolist(),
                "num_motions": n}
