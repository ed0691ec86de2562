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
  * the adversarial-motion-prior path (rows a12 / a24, ``enable_disc``): AMP observation windows recorded per step
    (:377), discriminator reward -log(max(1 - sigmoid(D), 1e-4)) * scale combined 0.5 / 0.5 with the task reward
    (:1011-1041), demo / replay ring buffers with permuted sampling (replay_buffer.py:27-69, amp_agent.py:975-1057),
    and ``_disc_loss`` (:895-952) -- BCE on agent+replay vs demo logits, logit regulariser, weight decay and the
    gradient penalty, whose double backward is hand-derived in learning/disc.py -- added to the PPO loss with
    ``disc_coef``; ONE gradient-norm clip over policy + discriminator parameters, Adam on both flat buffers.
"""
import numpy as np
import os

import torch

from .. import kernels as K
from . import rlg
from .common_agent import CommonAgent
from .disc import DiscNetwork
from .graph import r4
from .running_mean_std import RunningMeanStd


class ReplayBuffer:
    """phc/learning/replay_buffer.py:27-69 on the device: a ring of rows with a permuted sampling order.
    ``sample_indices`` returns row indices instead of copies: consumers gather in their own kernels."""

    def __init__(self, buffer_size, width, device, generator=None):
        self._head, self._total_count, self._buffer_size = 0, 0, int(buffer_size)
        self._device = torch.device(device)
        self._gen = generator
        self.data = torch.zeros(self._buffer_size, width, dtype=torch.float32, device=self._device)
        self._sample_idx = torch.randperm(self._buffer_size, generator=generator).to(self._device)
        self._sample_head = 0

    def get_buffer_size(self):
        return self._buffer_size

    def get_total_count(self):
        return self._total_count

    def store(self, rows):
        n, size = rows.shape[0], self._buffer_size
        assert n <= size
        store_n = min(n, size - self._head)
        self.data[self._head:self._head + store_n, :rows.shape[1]] = rows[:store_n]
        if n - store_n > 0:
            self.data[0:n - store_n, :rows.shape[1]] = rows[store_n:]
        self._head = (self._head + n) % size
        self._total_count += n

    def sample_indices(self, n):
        size = self._buffer_size
        idx = torch.arange(self._sample_head, self._sample_head + n, device=self._device) % size
        rand_idx = self._sample_idx[idx]
        if self._total_count < size:
            rand_idx = rand_idx % self._head
        self._sample_head += n
        if self._sample_head >= size:
            self._sample_idx = torch.randperm(size, generator=self._gen).to(self._device)
            self._sample_head = 0
        return rand_idx

    def sample(self, n):
        return self.data[self.sample_indices(n)]


class AMPAgent(CommonAgent):
    def __init__(self, base_name, config):
        super().__init__(base_name, config)
        task = self.vec_env.env.task
        env_cfg = task.cfg.get("env", task.cfg)
        if config.get("use_seq_rl", False):
            self.dataset = rlg.AMPDataset(self.batch_size, self.minibatch_size, self.is_discrete, True, self.ppo_device, self.seq_len,
                                          generator=self.dataset.generator, permutation_device=self.dataset.permutation_device)
        self.save_kin_info = bool(env_cfg.get("save_kin_info", False))
        self.only_kin_loss = bool(env_cfg.get("only_kin_loss", False))
        self.temp_running_mean = bool(getattr(task, "temp_running_mean", True))
        self.kin_lr = float(getattr(task, "kin_lr", 5e-4))
        # the reference's AMPAgent always trains its discriminator (amp_agent.py:621-629, 704-709); here it follows the env: on whenever
        # the env exposes AMP observation windows, unless the config says otherwise
        self.enable_disc = bool(config.get("enable_disc", "amp_observation_space" in self.env_info))
        if self.enable_disc:
            self._load_amp_config(config)
        if self.save_kin_info:
            n = self.model.parameters_count()
            self.kin_exp_avg = torch.zeros(n, device=self.ppo_device)     # kin_optimizer = Adam(a2c_network.parameters(), kin_lr)
            self.kin_exp_avg_sq = torch.zeros(n, device=self.ppo_device)
            self.kin_step = 0
            self.kin_dict_info = None
        if getattr(task, "fitting", False):
            # amp_agent.py:70-76: distillation needs the TEACHER's observation statistics: load the normalisers of env.models[0], freeze
            # them, and take every network tensor whose name and shape match (load_my_state_dict, :27-33)
            if not getattr(task, "models_path", None):
                raise ValueError("env.fitting is set but env.models names no checkpoint to take the normaliser statistics from")
            checkpoint = torch.load(task.models_path[0], map_location=self.ppo_device)
            self.set_stats_weights(checkpoint)
            self.freeze_state_weights()
            self._load_matching_state(checkpoint["model"])
        self.running_mean_std_temp = self.running_mean_std.clone_frozen()
        self._disc_ring, self._disc_info_all, self._disc_pos = None, None, 0
        self.z_noise_provider = None                                      # tests inject the re-parameterisation noise
        self._kin_partials = None

    # ------------------------------------------------------------------ AMP discriminator (amp_agent.py:851-1057)
    def _load_amp_config(self, config):
        self._task_reward_w, self._disc_reward_w = float(config["task_reward_w"]), float(config["disc_reward_w"])
        self._amp_observation_space = self.env_info["amp_observation_space"]
        self._amp_dim = int(self._amp_observation_space.shape[0])
        self._amp_pitch = r4(self._amp_dim)
        self._amp_batch_size, self._amp_minibatch_size = int(config["amp_batch_size"]), int(config["amp_minibatch_size"])
        assert self._amp_minibatch_size <= self.minibatch_size
        self._disc_coef, self._disc_logit_reg = float(config["disc_coef"]), float(config["disc_logit_reg"])
        self._disc_grad_penalty, self._disc_weight_decay = float(config["disc_grad_penalty"]), float(config["disc_weight_decay"])
        self._disc_reward_scale = float(config["disc_reward_scale"])
        self._normalize_amp_input = bool(config.get("normalize_amp_input", True))
        if not self._normalize_amp_input or config.get("norm_disc_reward", False):
            raise NotImplementedError("normalize_amp_input: True / norm_disc_reward: False (the shipped settings)")
        self.disc = DiscNetwork(config["network"], self._amp_dim, device=self.ppo_device, split_k=int(config.get("split_k", 8)))
        self.disc.mixed_precision = self.mixed_precision
        # The discriminator chain runs on its own stream beside the actor / critic chain (_side_stream).  Round-6 finding (tools/mask_contend_probe.py,
        # DESIGN.md section 6): the fp32 x3 kernels' relu-grad epilogue that takes its derivative from the BIT MASK returns wrong values in one
        # 16-lane quarter of a wave now and then while ANOTHER stream's GEMM waves share the SIMD (0 differences alone, or reading the fp32
        # activations instead) -- the same class of hazard as round 3's packed-fp32 one.  A policy network that has a concurrent chain beside it
        # therefore keeps the activation re-read; the bf16-storage path's byte masks and single-chain agents (cfg2) are not affected.
        if self._side_stream() is not None and hasattr(self.model, "concurrent_chain"):
            self.model.concurrent_chain = True
        self._amp_input_mean_std = RunningMeanStd((self._amp_dim,), device=self.ppo_device)
        self.disc_exp_avg = torch.zeros(self.disc.n_flat, device=self.ppo_device)
        self.disc_exp_avg_sq = torch.zeros(self.disc.n_flat, device=self.ppo_device)
        self._amp_replay_keep_prob = float(config["amp_replay_keep_prob"])

    def _build_amp_buffers(self):
        gen = torch.Generator()
        gen.manual_seed(int(self.config.get("seed", 0)) + 77 + self.rank)
        self.experience_buffer.add("amp_obs", width=self._amp_dim, pitch=self._amp_pitch)
        self._amp_obs_demo_buffer = ReplayBuffer(int(self.config["amp_obs_demo_buffer_size"]), self._amp_pitch, self.ppo_device, gen)
        self._amp_replay_buffer = ReplayBuffer(int(self.config["amp_replay_buffer_size"]), self._amp_pitch, self.ppo_device, gen)
        self.tensor_list += ["amp_obs"]
        self._disc_r = torch.zeros(self.num_actors, self.horizon_length, 1, device=self.ppo_device)
        self._amp_norm_scratch = None

    def _fetch_amp_obs_demo(self, num_samples):
        return self.vec_env.env.fetch_amp_obs_demo(num_samples)

    def _init_amp_demo_buf(self):
        size = self._amp_obs_demo_buffer.get_buffer_size()
        for _ in range(int(np.ceil(size / self._amp_batch_size))):
            self._amp_obs_demo_buffer.store(self._fetch_amp_obs_demo(self._amp_batch_size))

    def _update_amp_demos(self):
        self._amp_obs_demo_buffer.store(self._fetch_amp_obs_demo(self._amp_batch_size))

    def _init_train(self):
        super()._init_train()
        if self.enable_disc and self._amp_obs_demo_buffer.get_total_count() == 0:
            self._init_amp_demo_buf()

    def _stat_modules(self):
        return super()._stat_modules() + ([self._amp_input_mean_std] if self.enable_disc else [])

    def set_eval(self):
        super().set_eval()
        if self.enable_disc:
            self._amp_input_mean_std.eval()

    def set_train(self):
        super().set_train()
        if self.enable_disc:
            self._amp_input_mean_std.train()

    def _calc_disc_rewards(self, amp_rows):
        """:1027-1041 on (B, amp_pitch) rows: disc_r = -log(max(1 - sigmoid(D(norm(x))), 1e-4)) * disc_reward_scale."""
        b = amp_rows.shape[0]
        chunk = min(b, 16384)
        if self._amp_norm_scratch is None or self._amp_norm_scratch.shape[0] != chunk:
            self._amp_norm_scratch = torch.zeros(chunk, self._amp_pitch, device=self.ppo_device)
        out = torch.empty(b, 1, device=self.ppo_device)
        for c0 in range(0, b, chunk):
            n = min(chunk, b - c0)
            xs = self._amp_norm_scratch[:n]
            self._amp_input_mean_std.forward(amp_rows[c0:c0 + n], out=xs, out_cols=self._amp_pitch, update=False)
            logits = self.disc.eval_disc(self._amp_norm_scratch)[:n]
            K.disc_reward(logits, n, self._disc_reward_scale, out[c0:c0 + n])
        return out

    def _rollout_rewards(self, td):
        if not self.enable_disc:
            return td["rewards"]
        eb = self.experience_buffer
        disc_r = self._calc_disc_rewards(eb.flat("amp_obs"))                     # env-major rows, same order as rewards' storage
        self._disc_r = disc_r.view(self.num_actors, self.horizon_length, 1)
        combined = self._task_reward_w * eb.phys["rewards"] + self._disc_reward_w * self._disc_r      # _combine_rewards (:1011-1016)
        self._mb_rewards = combined
        return combined.transpose(0, 1)                                          # (T, N, 1) view like td["rewards"]

    def _store_replay_amp_obs(self, amp_obs):
        """:1043-1057."""
        buf = self._amp_replay_buffer
        if buf.get_total_count() > buf.get_buffer_size():
            keep = torch.bernoulli(torch.full((amp_obs.shape[0],), self._amp_replay_keep_prob, device=self.ppo_device)) == 1.0
            amp_obs = amp_obs[keep]
        if amp_obs.shape[0] > buf.get_buffer_size():
            amp_obs = amp_obs[torch.randperm(amp_obs.shape[0], device=self.ppo_device)[:buf.get_buffer_size()]]
        buf.store(amp_obs)

    def _param_groups(self):
        groups = super()._param_groups()
        if self.enable_disc:
            groups.append((self.disc.flat, self.disc.grad, self.disc_exp_avg, self.disc_exp_avg_sq, self.disc.n_flat))
        return groups

    def _side_stream(self):
        # The discriminator chain reads the dataset and its own parameters / normaliser, writes its own gradient slabs, statistics row and slice
        # of the norm partials: independent of the actor / critic chain until _apply_gradients.  PULSE_DISC_STREAM=0: inline (A/B switch).
        if not self.enable_disc or not str(self.ppo_device).startswith("cuda") or os.environ.get("PULSE_DISC_STREAM", "1") == "0":
            return None
        if getattr(self, "_disc_stream", None) is None:
            self._disc_stream = torch.cuda.Stream(device=self.ppo_device)
            self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
        return self._disc_stream

    def _extra_gradients(self, input_dict, idx):
        """The discriminator part of AMPAgent.calc_gradients (:621-629, 700-712): the first amp_minibatch_size rows of
        the minibatch for agent / replay / demo observations, normalised in that order (each call also updates the
        running statistics), then _disc_loss and its gradients."""
        if not self.enable_disc:
            return {}
        b = self._amp_minibatch_size
        d = input_dict["dataset"]
        sub = idx[:b]
        ws = self.disc.workspace(b)
        eb_rows = d["_amp_store"]
        X = ws["X"]
        rms = self._amp_input_mean_std
        rms.forward(eb_rows, row_idx=sub, out=X[0:b], out_cols=self._amp_pitch)                                   # amp_obs
        rep_src, rep_idx = d["_amp_replay_src"], d["_amp_replay_idx"]
        if rep_idx is not None and "_amp_idx_pair" in d:
            both = d["_amp_idx_pair"][:, sub]                              # replay / demo ring rows of this minibatch in ONE gather launch
            rep_rows, demo_rows = both[0], both[1]
        else:
            rep_rows, demo_rows = (rep_idx[sub] if rep_idx is not None else sub), d["_amp_demo_idx"][sub]
        rms.forward(rep_src, row_idx=rep_rows, out=X[b:2 * b], out_cols=self._amp_pitch)                          # amp_obs_replay
        rms.forward(self._amp_obs_demo_buffer.data, row_idx=demo_rows, out=X[2 * b:3 * b], out_cols=self._amp_pitch)  # amp_obs_demo
        logits = self.disc.forward(ws)
        # prediction loss, its logit gradients, accuracies and logit means in ONE launch (pulse_disc_head); everything the reported losses
        # need lands in one 20-float row: [disc_head's 8 | sum ||dD/dx||^2 at 8 | the reduce's eight per-region sums of squared parameters at 9..16:
        # ||W1||^2 at 9, ||W2||^2 at 11, ||w3||^2 at 13, the bias regions' in between]
        lazy = self._lazy_info and self._disc_ring is not None and self._disc_pos < self._disc_ring.shape[0]
        row = self._disc_ring[self._disc_pos] if lazy else torch.empty(20, dtype=torch.float32, device=self.ppo_device)
        scale = self._disc_coef / self.world_size
        if ws["b16"]:
            self.disc.loss_head_b16(ws, logits, b, scale, row[:8])
        else:
            K.disc_head(logits, b, scale, ws["dlogits"], row[:8])
        sqp = None
        if self._sq_fuse:                                          # inside calc_gradients on one GPU: the reduce launch also leaves the norm clip's sums of squares (group 1)
            sqp = self._sq_slice(1)
            self._sq_done.add(1)
        self.disc.backward(ws, self._disc_grad_penalty, self._disc_logit_reg, self._disc_weight_decay, scale=scale, stats=row[8:17], sq_partials=sqp)
        if lazy:                                                   # reduced once at the end of train_epoch (_end_loss_ring)
            out = self._disc_info_all[self._disc_pos]
            self._disc_pos += 1
        else:
            out = self._disc_info(row.unsqueeze(0), b)[0]
        return {"disc_loss": out[0], "disc_grad_penalty": out[1], "disc_logit_loss": out[2],
                "disc_agent_acc": out[3], "disc_demo_acc": out[4], "disc_agent_logit": out[5], "disc_demo_logit": out[6]}

    def _disc_info(self, raw, b, out=None):
        """(n, 20) raw rows [disc_head's 8 | sum ||dD/dx||^2 | per-region sums of squared parameters: ||W1||^2, b1, ||W2||^2, b2, ||w3||^2, b3, 0, 0]
        -> (n, 7): disc_loss (:895-952: prediction loss + logit regulariser + gradient penalty + weight decay), penalty, logit loss,
        accuracies, logit means."""
        out = torch.empty(raw.shape[0], 7, device=raw.device) if out is None else out
        pen, w1, w2, w3 = raw[:, 8] / b, raw[:, 9], raw[:, 11], raw[:, 13]
        out[:, 0] = raw[:, 0] + self._disc_logit_reg * w3 + self._disc_grad_penalty * pen
        if self._disc_weight_decay != 0:
            out[:, 0] += self._disc_weight_decay * (w1 + w2 + w3)
        out[:, 1], out[:, 2] = pen, w3
        out[:, 3:7] = raw[:, 3:7]
        return out

    def _begin_loss_ring(self, slots):
        super()._begin_loss_ring(slots)
        self._disc_pos = 0
        if self.enable_disc and self._lazy_info:
            if self._disc_ring is None or self._disc_ring.shape[0] < slots:
                self._disc_ring = torch.zeros(slots, 20, device=self.ppo_device)
            self._disc_info_all = torch.zeros(slots, 7, device=self.ppo_device)     # fresh per epoch: last epoch's dicts stay valid

    def _end_loss_ring(self):
        if self.enable_disc and self._lazy_info and self._disc_pos:
            n = self._disc_pos
            self._disc_info(self._disc_ring[:n], self._amp_minibatch_size, out=self._disc_info_all[:n])
        super()._end_loss_ring()

    # ------------------------------------------------------------------ checkpoint surface (amp_agent.py:81-118, 181-190)
    def get_stats_weights(self):
        state = {}
        if self.normalize_input:
            state["running_mean_std"] = self.running_mean_std.state_dict()
        if self.normalize_value:
            state["reward_mean_std"] = self.value_mean_std.state_dict()
        if self.enable_disc:
            state["amp_input_mean_std"] = self._amp_input_mean_std.state_dict()
        return state

    def set_stats_weights(self, weights):
        if self.normalize_input and weights["running_mean_std"]["running_mean"].shape == self.running_mean_std.running_mean.shape:
            self.running_mean_std.load_state_dict(weights["running_mean_std"])
        if self.normalize_value and "reward_mean_std" in weights:
            self.value_mean_std.load_state_dict(weights["reward_mean_std"])
        if self.enable_disc and "amp_input_mean_std" in weights:
            if weights["amp_input_mean_std"]["running_mean"].shape == self._amp_input_mean_std.running_mean.shape:
                self._amp_input_mean_std.load_state_dict(weights["amp_input_mean_std"])

    def freeze_state_weights(self):
        """amp_agent.py:123-131."""
        if self.normalize_input:
            self.running_mean_std.freeze()
        if self.normalize_value:
            self.value_mean_std.freeze()
        if self.mixed_precision:
            raise NotImplementedError("the reference raises for fitting with mixed_precision (amp_agent.py:130-131)")

    def unfreeze_state_weights(self):
        """amp_agent.py:133-141."""
        if self.normalize_input:
            self.running_mean_std.unfreeze()
        if self.normalize_value:
            self.value_mean_std.unfreeze()

    def _load_matching_state(self, saved):
        """load_my_state_dict (amp_agent.py:27-33): copy the saved tensors whose name exists here with the same shape; skip the rest."""
        own = self.model.state_dict()
        own.update({k: v for k, v in saved.items() if k in own and tuple(own[k].shape) == tuple(v.shape)})
        self.model.load_state_dict(own)
        if self.enable_disc:
            own = self.disc.state_dict()
            own.update({k: v for k, v in saved.items() if k in own and tuple(own[k].shape) == tuple(v.shape)})
            self.disc.load_state_dict(own)

    def get_full_state_weights(self):
        """The reference's checkpoint dict: 'model' holds EVERY a2c_network.* tensor (policy, critic, discriminator) under
        the reference's names, the normalisers sit next to it.  The Adam moments are stored as flat buffers (the reference
        stores torch.optim's index-keyed state, whose parameter order belongs to rl_games' module tree)."""
        state = super().get_full_state_weights()
        if self.enable_disc:
            state["model"].update(self.disc.state_dict())
            state["amp_input_mean_std"] = self._amp_input_mean_std.state_dict()
            state["disc_optimizer"] = {"exp_avg": self.disc_exp_avg.clone(), "exp_avg_sq": self.disc_exp_avg_sq.clone()}
        if self.save_kin_info:
            state["kin_optimizer"] = {"exp_avg": self.kin_exp_avg.clone(), "exp_avg_sq": self.kin_exp_avg_sq.clone(), "step": self.kin_step}
        return state

    def set_full_state_weights(self, weights):
        model = dict(weights["model"])
        if self.enable_disc:
            disc_keys = [k for k in model if "._disc_" in k]
            self.disc.load_state_dict({k: model.pop(k) for k in disc_keys})
            if "amp_input_mean_std" in weights:
                self._amp_input_mean_std.load_state_dict(weights["amp_input_mean_std"])
            if "disc_optimizer" in weights:
                self.disc_exp_avg.copy_(weights["disc_optimizer"]["exp_avg"])
                self.disc_exp_avg_sq.copy_(weights["disc_optimizer"]["exp_avg_sq"])
        else:
            model = {k: v for k, v in model.items() if "._disc_" not in k}
        super().set_full_state_weights(dict(weights, model=model))
        if self.save_kin_info and "kin_optimizer" in weights and "exp_avg" in weights["kin_optimizer"]:
            self.kin_exp_avg.copy_(weights["kin_optimizer"]["exp_avg"])
            self.kin_exp_avg_sq.copy_(weights["kin_optimizer"]["exp_avg_sq"])
            self.kin_step = int(weights["kin_optimizer"]["step"])
        self.running_mean_std_temp = self.running_mean_std.clone_frozen()

    # ------------------------------------------------------------------ epoch hooks (amp_agent.py:557-583)
    def pre_epoch(self, epoch_num):
        """amp_agent.py:557-579: motions re-drawn every shape_resampling_interval epochs, the get-up schedule (which also switches the
        task / discriminator reward weights), then the frozen copy of the observation normaliser."""
        task = self.vec_env.env.task
        interval = int(getattr(task, "shape_resampling_interval", 0) or 0)
        if interval > 0 and epoch_num > 1 and epoch_num % interval == 1 and hasattr(task, "resample_motions"):
            task.resample_motions()
        if getattr(task, "getup_schedule", False):
            task.update_getup_schedule(epoch_num, getup_udpate_epoch=task.getup_udpate_epoch)
            if epoch_num > task.getup_udpate_epoch:
                self._task_reward_w, self._disc_reward_w = 0.5, 0.5
            else:
                self._task_reward_w, self._disc_reward_w = 0.0, 1.0
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
        if self.enable_disc:
            self._build_amp_buffers()
            if self._amp_obs_demo_buffer.get_total_count() == 0:
                self._init_amp_demo_buf()

    def train_epoch(self):
        self.pre_epoch(self.epoch_num)
        info = super().train_epoch()
        if self.enable_disc:
            self._store_replay_amp_obs(self.experience_buffer.flat("amp_obs"))       # :534
            info["disc_rewards"] = self._disc_r
            info["mb_rewards"] = self._mb_rewards
        self.post_epoch(self.epoch_num)
        return info

    # ------------------------------------------------------------------ rollout additions (amp_agent.py:341-439)
    def _action_for_env(self, res_dict):
        return res_dict["mus"] if (self.only_kin_loss and self.save_kin_info) else res_dict["actions"]

    def _before_env_step(self, n):
        if self.enable_disc:
            # the env writes the step's finished AMP window straight into experience-buffer slot n (no (N, 1960) copy: update_data sees its own row)
            task = self.vec_env.env.task
            if hasattr(task, "set_amp_obs_sink"):
                task.set_amp_obs_sink(self.experience_buffer.slot("amp_obs", n))

    def _after_env_step(self, n, infos):
        if self.enable_disc:
            self.experience_buffer.update_data("amp_obs", n, infos["amp_obs"])          # :377
        if self.save_kin_info:
            flat = torch.cat([v.reshape(v.shape[0], -1).float() for v in infos["kin_dict"].values()], dim=-1)
            self.experience_buffer.update_data("kin_dict", n, flat)

    def prepare_dataset(self, batch_dict):
        d = super().prepare_dataset(batch_dict)
        if self.save_kin_info:
            d["kin_dict"] = batch_dict["kin_dict"]
        if self.enable_disc:
            # amp_agent.py:474-484: a fresh demo batch enters the demo ring, then T*N demo / replay rows are drawn.  Only the
            # first amp_minibatch_size rows of every minibatch are ever used (:621-628), so the draws are kept as ROW INDICES
            # into the rings and gathered inside the normaliser kernel instead of materialising two (T*N, 1960) copies.
            self._update_amp_demos()
            n = self.batch_size
            d["_amp_store"] = self.experience_buffer.flat("amp_obs")
            d["_amp_demo_idx"] = self._amp_obs_demo_buffer.sample_indices(n)
            if self._amp_replay_buffer.get_total_count() == 0:
                d["_amp_replay_src"], d["_amp_replay_idx"] = d["_amp_store"], None      # batch_dict['amp_obs_replay'] = batch_dict['amp_obs']
            else:
                d["_amp_replay_src"], d["_amp_replay_idx"] = self._amp_replay_buffer.data, self._amp_replay_buffer.sample_indices(n)
                d["_amp_idx_pair"] = torch.stack((d["_amp_replay_idx"], d["_amp_demo_idx"]))
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
        """amp_agent.py:771-849 (z_type 'vae', learned prior): action RMSE + kld_coefficient KL(q || prior) + ar1_coefficient AR(1) latent
        smoothness (seams masked through progress_buf) + 0.005 regulariser, own Adam(kin_lr).  Head algebra: pulse_vae_kin_loss (losses +
        d loss / d pred_action) and pulse_vae_head_backward (encoder / prior head gradients); no autograd."""
        task = self.vec_env.env.task
        model = self.model
        net = model.net
        gt_action = kin_dict["gt_action"]
        if self.z_noise_provider is not None:
            ws["z_noise"] = self.z_noise_provider(mb)
        model.forward_actor(ws, mb)
        model.compute_prior(ws)
        g = ws["g"]
        E, t = net.embedding_size, self.horizon_length
        use_ar1, use_regu = bool(getattr(task, "use_ar1_prior", False)), bool(getattr(task, "use_vae_prior_regu", False))
        # (the experience buffer stores kin_dict as floats: progress values are small integers, exact either way)
        prog = kin_dict["progress_buf"].reshape(-1).to(torch.int64).contiguous() if use_ar1 else None
        if self._kin_partials is None:
            self._kin_partials = torch.zeros(256, 8, device=self.ppo_device)
        gt = gt_action if gt_action.stride(-1) == 1 else gt_action.contiguous()
        K.vae_kin_loss(ws["mu"], gt, g.act_bufs["zheads"], g.act_bufs["pheads"], prog, ws["dmu"], self._kin_partials, rows=mb,
                       num_actions=self.actions_num, embedding_size=E, horizon=t, clamp=net.use_vae_clamped_prior, clamp_max=net.vae_var_clamp_max,
                       use_ar1=use_ar1, use_regu=use_regu)
        # the loss terms from the per-workgroup partials in three small launches (fixed-order column sum, one scaling, one dot product) instead of
        # a dozen scalar ops: sums = [sum sq action error -> RMSE term, KL, AR(1), four regulariser sums, -]
        n_err = (mb // t) * (t - 1)
        kld_w = float(task.kld_coefficient)
        key = (mb, t, E, use_ar1, use_regu, kld_w, float(task.ar1_coefficient))
        if getattr(self, "_kin_scale_key", None) != key:
            r = 0.001 / (mb * E) if use_regu else 0.0
            self._kin_scale = torch.tensor([1.0 / mb, 1.0 / mb, (1.0 / n_err) if use_ar1 else 0.0, r, r, r, r, 0.0], device=self.ppo_device)
            self._kin_weight = torch.tensor([1.0, kld_w, float(task.ar1_coefficient), 0.005, 0.005, 0.005, 0.005, 0.0], device=self.ppo_device)
            self._kin_scale_key = key
        terms = self._kin_partials.sum(0) * self._kin_scale
        info = {}
        kin_action_loss, kld = terms[0], terms[1]
        if use_ar1:
            info["kin_ar1"] = terms[2]
        if use_regu:
            info["kin_prior_regu"] = terms[3:7].sum()
        kin_loss = torch.dot(terms, self._kin_weight)
        info["kin_action_loss"], info["kin_KLD"] = kin_action_loss, kld
        if task.kld_anneal:                                                     # :826-832 (after the loss: this step used the old weight)
            if self.epoch_num > 2500:
                mn = task.kld_coefficient_min
                task.kld_coefficient = (0.01 - mn) * max((5000 - self.epoch_num) / 2500, 0) + mn
            info["kin_kld_w"] = task.kld_coefficient
        # ---- backward through the GEMM plans; the head-level gradients join at the encoder / prior heads
        model.backward_actor(ws, kin={"c_kl": kld_w / mb, "c_ar1": (task.ar1_coefficient / n_err) if use_ar1 else 0.0,
                                      "c_regu": (0.005 * 0.001 / (mb * E)) if use_regu else 0.0, "progress": prog, "horizon": t})
        model.backward_prior(ws)
        # the critic (PPO pass only) is not visited: its gradient is written as zeros by the region reduce, which also leaves the norm clip's
        # sums of squares on one GPU (with data parallelism the norm is taken after the all-reduce)
        sqp = self._sq_slice(0)
        fused = model.book.reduce_grads(1.0 / self.world_size, untouched=model._untouched(ws, ("dec", "enc", "prior")),
                                        sq_partials=None if self.multi_gpu else sqp)
        if self.multi_gpu:
            self.dist.sync_gradients(model.grad)
        self.kin_step += 1
        if not fused:
            K.sqnorm_partial(model.grad, model.n_flat, sqp)
        K.adam_step(model.flat, model.grad, self.kin_exp_avg, self.kin_exp_avg_sq, model.n_flat, lr=self.kin_lr, step=self.kin_step,
                    max_norm=self.grad_norm, sqnorm_partials=sqp, grad_norm_out=self._grad_norm)
        info["kin_loss"] = kin_loss
        info["grad_norm"] = self._grad_norm.clone()
        return info
