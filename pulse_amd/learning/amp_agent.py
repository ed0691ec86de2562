"""AMPAgent on MI355X -- the pieces of phc/learning/amp_agent.py that PULSE training uses.

Built this round:
  * frozen-statistics observation normaliser: ``pre_epoch`` deep-copies + freezes the input normaliser, the
    network is fed observations normalised by that copy while the LIVE statistics keep being updated on every
    minibatch (amp_agent.py:557-603) -- here ONE kernel pass does both;
  * ``only_kin_loss`` mode (PULSE distillation, env_im_vae.yaml:51): the env is stepped with ``mus``
    (:367-371), ``kin_dict`` is recorded into the experience buffer (:379-383), minibatches are whole env
    sequences (``use_seq_rl``, :40-42) and ``calc_gradients`` runs ``_optimize_kin`` (:771-849): action RMSE +
    beta KL(q || learned prior) + AR(1) latent smoothness (+ optional regulariser), its own Adam
    (``kin_optimizer``, lr ``kin_lr``), clip 50, KL-weight annealing over epochs 2500-5000.
NOT built yet (raise if enabled): the adversarial discriminator path (_disc_loss / AMP observations / replay
buffers, rows a12 / a24).
"""
import torch

from .. import kernels as K
from . import rlg
from .common_agent import CommonAgent


def kl_multi(qm, qv, pm, pv):
    """phc/learning/loss_functions.py:3-10."""
    return (0.5 * (pv - qv + qv.exp() / pv.exp() + (qm - pm).pow(2) / pv.exp() - 1)).sum(-1)


class AMPAgent(CommonAgent):
    def __init__(self, base_name, config):
        super().__init__(base_name, config)
        task = self.vec_env.env.task
        env_cfg = task.cfg.get("env", task.cfg)
        if config.get("use_seq_rl", False):
            self.dataset = rlg.AMPDataset(self.batch_size, self.minibatch_size, self.is_discrete, True, self.ppo_device, self.seq_len,
                                          generator=self.dataset.generator)
        self.save_kin_info = bool(env_cfg.get("save_kin_info", False))
        self.only_kin_loss = bool(env_cfg.get("only_kin_loss", False))
        self.temp_running_mean = bool(getattr(task, "temp_running_mean", True))
        self.kin_lr = float(getattr(task, "kin_lr", 5e-4))
        if config.get("enable_disc", False):
            raise NotImplementedError("AMP discriminator training (rows a12 / a24) is not built yet")
        if self.save_kin_info:
            n = self.model.parameters_count()
            self.kin_exp_avg = torch.zeros(n, device=self.ppo_device)     # kin_optimizer = Adam(a2c_network.parameters(), kin_lr)
            self.kin_exp_avg_sq = torch.zeros(n, device=self.ppo_device)
            self.kin_step = 0
            self.kin_dict_info = None
        self.running_mean_std_temp = self.running_mean_std.clone_frozen()
        self.z_noise_provider = None                                      # tests inject the re-parameterisation noise

    # ------------------------------------------------------------------ epoch hooks (amp_agent.py:557-583)
    def pre_epoch(self, epoch_num):
        self.running_mean_std_temp = self.running_mean_std.clone_frozen()

    def post_epoch(self, epoch_num):
        self.running_mean_std_temp = self.running_mean_std.clone_frozen()

    def _obs_normalizer_for_update(self):
        if self.temp_running_mean:
            return self.running_mean_std_temp, self.running_mean_std
        return self.running_mean_std, None

    def init_tensors(self):
        super().init_tensors()
        if self.save_kin_info:
            kd = self.vec_env.env.task.kin_dict
            self.kin_dict_info = {k: (v.shape, v.reshape(v.shape[0], -1).shape) for k, v in kd.items()}
            self.kin_dict_size = sum(v.reshape(v.shape[0], -1).shape[-1] for v in kd.values())
            self.experience_buffer.add("kin_dict", width=self.kin_dict_size)
            self.tensor_list += ["kin_dict"]

    def train_epoch(self):
        self.pre_epoch(self.epoch_num)
        info = super().train_epoch()
        self.post_epoch(self.epoch_num)
        return info

    # ------------------------------------------------------------------ rollout additions (amp_agent.py:341-439)
    def _action_for_env(self, res_dict):
        return res_dict["mus"] if (self.only_kin_loss and self.save_kin_info) else res_dict["actions"]

    def _after_env_step(self, n, infos):
        if self.save_kin_info:
            flat = torch.cat([v.reshape(v.shape[0], -1).float() for v in infos["kin_dict"].values()], dim=-1)
            self.experience_buffer.update_data("kin_dict", n, flat)

    def prepare_dataset(self, batch_dict):
        d = super().prepare_dataset(batch_dict)
        if self.save_kin_info:
            d["kin_dict"] = batch_dict["kin_dict"]
        self.dataset.update_values_dict(d, rnn_format=True, horizon_length=self.horizon_length, num_envs=self.num_actors)
        return d

    def _assamble_kin_dict(self, kin_dict_flat):
        b, acc, out = kin_dict_flat.shape[0], 0, {}
        for k, v in self.kin_dict_info.items():
            out[k] = kin_dict_flat[:, acc:acc + v[1][-1]].view(b, *v[0][1:])
            acc += v[1][-1]
        return out

    # ------------------------------------------------------------------ update
    def calc_gradients(self, input_dict):
        if not self.only_kin_loss:
            return super().calc_gradients(input_dict)
        self.set_train()
        idx, obs_store = input_dict["idx"], input_dict["dataset"]["_obs_store"]
        mb = idx.numel()
        ws = self.model.workspace(mb, train=True)
        norm, live = self._obs_normalizer_for_update()
        if live is not None:
            live.forward(obs_store, row_idx=idx, out=ws["x"], out_cols=self.model.in_pitch, norm_with=norm)
        else:
            norm.forward(obs_store, row_idx=idx, out=ws["x"], out_cols=self.model.in_pitch)
        kin = self._assamble_kin_dict(input_dict["dataset"]["kin_dict"][idx])
        info = self._optimize_kin(ws, mb, kin)
        zero = torch.zeros((), device=self.ppo_device)
        self.train_result = {"entropy": zero, "kl": zero, "last_lr": self.last_lr, "lr_mul": 0.0}
        self.train_result.update(info)

    def _optimize_kin(self, ws, mb, kin_dict):
        """amp_agent.py:771-849 (z_type 'vae', learned prior)."""
        task = self.vec_env.env.task
        model = self.model
        gt_action = kin_dict["gt_action"]
        if self.z_noise_provider is not None:
            ws["z_noise"] = self.z_noise_provider(mb)
        model.forward_actor(ws, mb, need_grad=True)
        prior_mu, prior_log_var = model.compute_prior(ws, need_grad=True)
        info = {}
        with torch.enable_grad():
            pred_action = ws["mu"].detach().clone().requires_grad_(True)
            vae_mu, vae_log_var = ws["vae_mu"], ws["vae_log_var"]
            kin_action_loss = torch.norm(pred_action - gt_action, dim=-1).mean()
            kld = kl_multi(vae_mu, vae_log_var, prior_mu, prior_log_var).mean()
            ar1_prior, regu_prior = 0, 0
            if getattr(task, "use_ar1_prior", False):
                t = self.horizon_length
                time_zs = vae_mu.view(mb // t, t, -1)
                error = time_zs[:, 1:] - time_zs[:, :-1] * 0.99
                idxes = kin_dict["progress_buf"].view(mb // t, t, -1)
                not_consecs = ((idxes[:, 1:] - idxes[:, :-1]) != 1).view(-1)
                starteres = ((idxes <= 2)[:, 1:] + (idxes <= 2)[:, :-1]).view(-1)
                keep = (~(not_consecs | starteres)).to(error.dtype)
                error = error.reshape(-1, error.shape[-1]) * keep[:, None]      # error[mask] = 0
                ar1_prior = torch.norm(error, dim=-1).mean()
                info["kin_ar1"] = ar1_prior.detach()
            if getattr(task, "use_vae_prior_regu", False):
                regu_prior = ((prior_mu ** 2).mean() + (vae_mu ** 2).mean()) * 0.001 + ((prior_log_var ** 2).mean() + (vae_log_var ** 2).mean()) * 0.001
                info["kin_prior_regu"] = regu_prior.detach()
            kin_loss = kin_action_loss + kld * task.kld_coefficient + ar1_prior * task.ar1_coefficient + regu_prior * 0.005
            g_pred, = torch.autograd.grad(kin_loss, pred_action, retain_graph=True)
        info["kin_action_loss"], info["kin_KLD"] = kin_action_loss.detach(), kld.detach()
        if task.kld_anneal:                                                     # :826-832
            if self.epoch_num > 2500:
                mn = task.kld_coefficient_min
                task.kld_coefficient = (0.01 - mn) * max((5000 - self.epoch_num) / 2500, 0) + mn
            info["kin_kld_w"] = task.kld_coefficient
        # ---- backward through the GEMM plans; head-level gradients come from the autograd graph above
        model.book.slabs.zero_()
        ws["dmu"].copy_(g_pred)
        model.backward_actor(ws, extra_loss=kin_loss)
        model.backward_prior(ws)
        model.book.reduce_grads(1.0 / self.world_size)
        if self.multi_gpu:
            self.dist.sync_gradients(model.grad)
        self.kin_step += 1
        K.sqnorm_partial(model.grad, model.n_flat, self._sq_partials)
        K.adam_step(model.flat, model.grad, self.kin_exp_avg, self.kin_exp_avg_sq, model.n_flat, lr=self.kin_lr, step=self.kin_step,
                    max_norm=self.grad_norm, sqnorm_partials=self._sq_partials, grad_norm_out=self._grad_norm)
        info["kin_loss"] = kin_loss.detach()
        info["grad_norm"] = self._grad_norm.clone()
        return info
