"""HumanoidZ on MI355X: the frozen PULSE prior + decoder INSIDE ``env.step`` (downstream-task mode).

Mirrors phc/env/tasks/humanoid_z.py:
  initialize_z_models   :24-67   load decoder / prior / running stats from a PULSE checkpoint
                                  (phc/learning/network_loader.py:139-176 reads them by key prefix; the
                                  checkpoint produced by pulse_amd's AMPZNetwork.state_dict() uses the same keys)
  compute_z_actions     :81-155  self_obs = (obs[:, :358] - mean) / sqrt(var + 1e-5)   [NOT clamped for the prior]
                                  z = z_prior_mu(z_prior(self_obs)) + action_z           [use_vae_prior]
                                  action = decoder(cat(clamp(self_obs, +-5), z))
  step_z                :157-177 latent action -> PD action -> the ordinary step phases
Both MLPs run as forward launch plans of the fp32 MFMA GEMM (network_z.AMPZNetwork "prior" and "dec" plans);
the two normalisations are the pulse_rms_normalize kernel (clip = inf for the prior input, 5 for the decoder).
"""
import torch

from .. import kernels as K
from ..learning.network_z import AMPZNetwork
from .humanoid_im import HumanoidIm


class HumanoidZ:
    def initialize_z_models(self, checkpoint, net_params, task_obs_size_detail=None):
        """checkpoint: {'model': state_dict with reference key names, 'running_mean_std': {...}}."""
        detail = {"embedding_size": self._embedding_size, "z_type": "vae", "use_vae_prior": True, "use_vae_clamped_prior": True,
                  "vae_var_clamp_max": 2}
        detail.update(task_obs_size_detail or {})
        self._z_net = AMPZNetwork(net_params, actions_num=self._dof_size, self_obs_size=self.get_self_obs_size(),
                                  task_obs_size=detail.get("distill_task_obs_size", 576), task_obs_size_detail=detail, device=self.device)
        self._z_net.load_state_dict(checkpoint["model"])
        rms = checkpoint["running_mean_std"]
        self.running_mean = rms["running_mean"].to(self.device, torch.float64).contiguous()
        self.running_var = rms["running_var"].to(self.device, torch.float64).contiguous()
        self._z_graph = self._z_net.graph(self.num_envs)

    def compute_z_actions(self, action_z):
        net, G = self._z_net, self._z_graph
        g = G["g"]
        n, s, zc, e = self.num_envs, self.get_self_obs_size(), net.z_col, net.embedding_size
        obs = self._obs_store
        # prior input: normalised, unclamped self observation (humanoid_z.py:87)
        K.rms_normalize(obs, self.running_mean, self.running_var, rows=n, cols=s, x_stride=obs.stride(0), y=G["x"], y_stride=G["x"].stride(0),
                        y_cols=s, clip=3.0e38)
        G["fwd_prior"].run()
        prior_mu = g.act_bufs["pheads"][:, :e]
        ain = g.act_bufs["ain"]
        # decoder input: clamp(self_obs, +-5) (:149) | z = prior_mu + action_z (:102-103); project_to_norm(.., "none") is a no-op
        K.rms_normalize(obs, self.running_mean, self.running_var, rows=n, cols=s, x_stride=obs.stride(0), y=ain, y_stride=ain.stride(0),
                        y_cols=s, clip=5.0)
        torch.add(prior_mu, action_z, out=ain[:, zc:zc + e])
        G["fwd_dec"].run()
        return g.act_bufs["mu"][:, :self._dof_size]

    def step_z(self, action_z):
        self.action_z = action_z
        actions = self.compute_z_actions(action_z)
        self.pre_physics_step(actions)
        self._physics_step()
        self.post_physics_step()


class HumanoidImZ(HumanoidIm, HumanoidZ):
    """HumanoidImZ (phc/env/tasks/humanoid_im.py:1199-1214): imitation task whose action is the 32-d latent."""

    def __init__(self, cfg, sim, motion_lib, device="cuda:0"):
        super().__init__(cfg, sim, motion_lib, device=device)
        env = cfg.get("env", cfg)
        self._embedding_size = int(env.get("embedding_size", 32))
        self.num_actions = self._embedding_size                                # _setup_character_props_z (:65-67)

    def step(self, action_z):
        self.step_z(action_z)
