"""CommonAgent on MI355X: PPO + GAE behind the rl_games-style ``train_epoch()`` / ``env.step()`` surface.

Mirrors phc/learning/common_agent.py (CommonAgent(a2c_continuous.A2CAgent)):
  __init__            :36-90      (+ the A2CBase config plumbing of rl_games 1.1.4)
  init_tensors        :92-98
  train               :100-185
  train_epoch         :191-260
  get_action_values   :262-288
  play_steps          :290-355
  prepare_dataset     :357-398
  calc_gradients      :400-491
  discount_values     :493-505
  bound_loss / _actor_loss / _critic_loss / _calc_advs   :512-520, :564-599
Method names, dict keys and tensor shapes follow the reference so the class drops into the
``algo_factory.register_builder(...)`` call of phc/run_hydra.py:250-268.

What is different underneath (MI355X-first):
  * every tensor op of the hot loops is a HIP kernel behind the C ABI (pulse_amd/csrc): the
    observation normaliser fused with the minibatch gather, fp32 MFMA GEMMs with fused
    bias/activation/derivative epilogues, one PPO loss+gradient kernel, one slab reduce, one fused
    clip+Adam over a flat parameter buffer.  No autograd graph, no per-parameter optimiser loop;
  * rollout storage is env-major so the (T,N)->(N*T) flatten is free; network outputs are written
    straight into their experience-buffer slots;
  * no device->host read inside an epoch (the reference syncs on `.nonzero()` every step and on
    `kl.item()` every minibatch); episode statistics use masked, device-side meters;
  * multi-GPU = one process per GPU; the single exchange step per optimiser step is one RCCL
    all-reduce of the flat gradient (pulse_amd/parallel.py).
"""
import math
import os
import time

import torch

from .. import kernels as K
from .. import ops
from ..parallel import DistContext
from . import rlg
from .network import A2CNetwork
from .running_mean_std import RunningMeanStd


def _record_stream(obj, stream):
    """tensor.record_stream(stream) for every CUDA tensor inside nested dicts / lists / tuples."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class CommonAgent:
    def __init__(self, base_name, config):
        # ---------------- A2CBase.__init__ (rl_games 3P) ----------------
        self.config = self.cfg = config
        self.base_name = base_name
        self.exp_name = str(config.get("train_dir", "output/pulse_amd")).split("/")[-1]
        self.nn_dir = config.get("network_path", str(config.get("train_dir", "output/pulse_amd")) + "/nn")
        self.multi_gpu = bool(config.get("multi_gpu", False))
        self.overlap_allreduce = bool(config.get("overlap_allreduce", True))   # gradient buckets all-reduced beside the backward
        self.dist = config.get("dist") or DistContext(enabled=None if self.multi_gpu else False)
        self.rank, self.world_size = self.dist.rank, self.dist.world_size
        self.ppo_device = self.device = torch.device(config.get("device", config.get("ppo_device", "cuda:0")))
        if self.ppo_device.type != "cuda":
            raise ValueError("pulse_amd agents run on the GPU only (there is no CPU path)")
        if "vec_env" in config:
            self.vec_env = config["vec_env"]
        else:                # A2CBase.__init__: vecenv.create_vec_env(env_name, num_actors, **env_config)
            from ..runner import create_vec_env
            self.vec_env = create_vec_env(config["env_name"], int(config["num_actors"]), **config.get("env_config", {}))
        self.env_info = config.get("env_info") or self.vec_env.get_env_info()
        self.num_actors = int(config.get("num_actors", self.vec_env.num_envs))
        self.num_agents = 1
        self.value_size = self.env_info.get("value_size", 1)
        self.observation_space = self.env_info["observation_space"]
        self.obs_shape = self.observation_space.shape
        self.normalize_input = bool(config.get("normalize_input", True))
        self.normalize_value = bool(config.get("normalize_value", True))
        self.normalize_advantage = bool(config.get("normalize_advantage", True))
        self.horizon_length = int(config["horizon_length"])
        self.gamma, self.tau = float(config["gamma"]), float(config["tau"])
        self.e_clip = float(config["e_clip"])
        self.clip_value = bool(config["clip_value"])
        self.critic_coef = float(config["critic_coef"])
        self.entropy_coef = float(config.get("entropy_coef", 0.0))
        self.grad_norm = float(config.get("grad_norm", 1.0))
        self.truncate_grads = bool(config.get("truncate_grads", False))
        self.mini_epochs_num = int(config["mini_epochs"])
        self.minibatch_size = int(config["minibatch_size"])
        self.batch_size = self.horizon_length * self.num_actors * self.num_agents
        assert self.batch_size % self.minibatch_size == 0, "batch_size must be divisible by minibatch_size"
        self.num_minibatches = self.batch_size // self.minibatch_size
        # mixed_precision: the reference wraps model forward + losses of calc_gradients in autocast (amp_agent.py:671,
        # common_agent.py:426) with a GradScaler.  Here it selects the bf16 MFMA for the TRAINING forward / backward GEMMs over
        # fp32 master weights (BASELINE.json configs[4] names bf16; bf16 needs no loss scaling); rollout inference stays fp32
        # like the reference's un-autocast get_action_values (common_agent.py:262-288).
        self.mixed_precision = bool(config.get("mixed_precision", False))
        self.weight_decay = float(config.get("weight_decay", 0.0))
        self.schedule_type = config.get("schedule_type", "legacy")
        self.is_adaptive_lr = config.get("lr_schedule", "constant") == "adaptive"
        self.scheduler = rlg.AdaptiveScheduler(config.get("kl_threshold", 0.008)) if self.is_adaptive_lr else rlg.IdentityScheduler()
        self.rewards_shaper = rlg.DefaultRewardsShaper(**config.get("reward_shaper", {}))
        self.max_epochs = int(config.get("max_epochs", 1e6))
        self.save_freq = int(config.get("save_frequency", 0))
        self.games_to_track = int(config.get("games_to_track", 100))
        self.seq_len = int(config.get("seq_length", 4))
        self.is_rnn = False
        self.has_central_value = False
        self.use_action_masks = False
        self.epoch_num = 0
        self.frame = 0
        self.curr_frames = 0
        self.states = None
        self.rnn_states = None

        # ---------------- CommonAgent.__init__ (common_agent.py:36-90) ----------------
        self._load_config_params(config)
        self.is_discrete = False
        self._setup_action_space()
        self.bounds_loss_coef = config.get("bounds_loss_coef", None)
        self.clip_actions = config.get("clip_actions", True)
        net_config = self._build_net_config()
        task = self.vec_env.env.task
        self.obs_pitch = getattr(task, "obs_pitch", (self.obs_shape[0] + 31) // 32 * 32)
        if self.normalize_input:
            self.running_mean_std = RunningMeanStd(task.get_running_mean_size(), device=self.ppo_device)
        else:
            raise NotImplementedError("normalize_input: False (every shipped config sets True; amp_agent.py:594-603 requires it)")
        self.value_mean_std = RunningMeanStd((1,), device=self.ppo_device) if self.normalize_value else None
        seed = int(config.get("seed", 0)) + self.rank                       # run_hydra.py:124
        self.noise_generator = torch.Generator(device=self.ppo_device)
        self.noise_generator.manual_seed(seed)
        self.model = self._build_model(net_config)
        if self.mixed_precision:
            if not hasattr(self.model, "mixed_precision"):
                raise NotImplementedError("mixed_precision is built for the actor / critic MLP and the discriminator (cfg5); the amp_z graph runs fp32")
            self.model.mixed_precision = True
        self.last_lr = float(self.last_lr)
        # Adam(lr, eps=1e-8) over the flat parameter buffer (common_agent.py:66)
        n = self.model.parameters_count()
        self.exp_avg = torch.zeros(n, device=self.ppo_device)
        self.exp_avg_sq = torch.zeros(n, device=self.ppo_device)
        self.optimizer_step = 0
        self._sq_partials = torch.zeros(256, device=self.ppo_device)
        self._meter_partials = None
        self._sq_done, self._sq_fuse = set(), False      # parameter groups whose sum-of-squares partials the gradient reduce of this step already produced
        self._grad_norm = torch.zeros(1, device=self.ppo_device)
        self._partials_ring, self._lazy_info, self._ring_pos = None, False, 0
        self._loss_partials = torch.zeros(max(1, min(1024, self.minibatch_size // 16)), 8, device=self.ppo_device)   # one pass of 16-sample groups per workgroup
        self._adv_partials = torch.zeros(128, 2, dtype=torch.float64, device=self.ppo_device)
        perm_gen = torch.Generator()
        perm_gen.manual_seed(seed)
        self.dataset = rlg.AMPDataset(self.batch_size, self.minibatch_size, self.is_discrete, self.is_rnn, self.ppo_device, self.seq_len,
                                      generator=perm_gen, permutation_device=config.get("permutation_device", "cuda"))
        self.game_rewards = rlg.AverageMeter((self.value_size,), self.games_to_track, self.ppo_device)
        self.game_lengths = rlg.AverageMeter((1,), self.games_to_track, self.ppo_device)
        self.train_result = {}
        self.noise_provider = None
        self.epoch_counter = 0
        self._entropy = None
        self._tensors_ready = False
        self._boot_idx = self._boot_val = None
        self._boot_shortcut = None           # decided on the first rollout (_bootstrap_values)
        # the rollout's two records of every observation (obses[n], next_obses[n]) written by the kernels that hold the row anyway -- the
        # normaliser pass and the env's step kernel -- instead of by two 15 MB copy launches per step; PULSE_OBS_SINK=0 restores the copies (A/B)
        self._obs_sink_enabled = os.environ.get("PULSE_OBS_SINK", "1") != "0"
        self._obs_fused_key, self._obs_fused_ok = None, False
        self._rollout_noise = None

    # ------------------------------------------------------------------ construction helpers
    def _load_config_params(self, config):
        self.last_lr = config["learning_rate"]

    def _setup_action_space(self):
        action_space = self.env_info["action_space"]
        self.actions_num = action_space.shape[0]
        self.actions_low = torch.from_numpy(action_space.low.copy()).float().to(self.ppo_device)
        self.actions_high = torch.from_numpy(action_space.high.copy()).float().to(self.ppo_device)

    def _build_net_config(self):
        return {"actions_num": self.actions_num, "input_shape": self.obs_shape, "num_seqs": self.num_actors * self.num_agents,
                "value_size": self.env_info.get("value_size", 1)}

    def _build_model(self, net_config):
        params = self.config["network"]
        if params.get("name", "amp") == "amp_z":
            from .model_z import AMPZModel
            task = self.vec_env.env.task
            return AMPZModel(params, actions_num=net_config["actions_num"], self_obs_size=task.get_self_obs_size(),
                             task_obs_size=task.get_task_obs_size(), task_obs_size_detail=task.get_task_obs_size_detail(),
                             device=self.ppo_device, split_k=int(self.config.get("split_k", 8)), generator=self.noise_generator)
        if params.get("name", "amp") == "amp_sept":
            from .network_sept import AMPSeptModel
            task = self.vec_env.env.task
            return AMPSeptModel(params, actions_num=net_config["actions_num"], self_obs_size=task.get_self_obs_size(),
                                task_obs_size=task.get_task_obs_size(), task_obs_size_detail=task.get_task_obs_size_detail(),
                                device=self.ppo_device, split_k=int(self.config.get("split_k", 8)))
        if params.get("name", "amp") == "amp_z_reader":
            # AMPZReaderBuilder.Network (amp_network_z_reader_builder.py:21-57) IS AMPBuilder.Network -- the plain actor / critic MLP
            # whose 32-d "action" is the latent a frozen PULSE decoder turns into joint targets inside env.step -- unless
            # vae_prior_policy swaps sigma for the decoder's prior log-variance (:46-55; no shipped config sets it)
            detail = self.vec_env.env.task.get_task_obs_size_detail() if hasattr(self.vec_env.env.task, "get_task_obs_size_detail") else {}
            if detail.get("vae_prior_policy", False):
                raise NotImplementedError("amp_z_reader with vae_prior_policy (sigma from the frozen prior)")
        return A2CNetwork(params, actions_num=net_config["actions_num"], input_shape=net_config["input_shape"],
                          value_size=net_config["value_size"], device=self.ppo_device, split_k=int(self.config.get("split_k", 8)))

    # ------------------------------------------------------------------ init_tensors (common_agent.py:92-98)
    def init_tensors(self):
        net = self.model
        self._claim_env_buffers()
        self.experience_buffer = rlg.ExperienceBuffer(self.num_actors, self.horizon_length, self.obs_shape[0], self.obs_pitch,
                                                      self.actions_num, net.a_pitch, self.ppo_device)
        self.experience_buffer.add("next_obses", like="obses")
        self.experience_buffer.add("next_values", like="values")
        self.experience_buffer.add("terminates", like="dones")
        self.current_rewards = torch.zeros(self.num_actors, self.value_size, device=self.ppo_device)
        self.current_lengths = torch.zeros(self.num_actors, device=self.ppo_device)
        self.dones = torch.ones(self.num_actors, dtype=torch.uint8, device=self.ppo_device)
        self._done_mask = torch.zeros(self.num_actors, dtype=torch.bool, device=self.ppo_device)
        sh = self.rewards_shaper
        if self.value_size != 1 or sh.min_val != -float("inf") or sh.max_val != float("inf"):
            raise NotImplementedError("rollout bookkeeping kernel: value_size 1, reward shaper without clamping (every shipped config)")
        self.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
        self.tensor_list = self.update_list + ["obses", "states", "dones"] + ["next_obses"]
        self._tensors_ready = True

    # ------------------------------------------------------------------ mode switches
    def set_eval(self):
        self.model.eval()
        self.running_mean_std.eval()
        if self.value_mean_std is not None:
            self.value_mean_std.eval()

    def set_train(self):
        self.model.train()
        self.running_mean_std.train()
        if self.value_mean_std is not None:
            self.value_mean_std.train()

    def update_lr(self, lr):
        self.last_lr = lr

    def update_epoch(self):
        self.epoch_num += 1
        return self.epoch_num

    # ------------------------------------------------------------------ env plumbing (A2CBase)
    def obs_to_tensors(self, obs):
        return obs if isinstance(obs, dict) else {"obs": obs}

    def env_reset(self, env_ids=None):
        obs = self.vec_env.reset(env_ids)
        return self.obs_to_tensors(obs)

    def _env_reset_masked(self, mask):
        return self.obs_to_tensors(self.vec_env.reset_masked(mask))

    def _claim_env_buffers(self):
        """This agent never holds a returned observation across env steps (it is copied into the experience buffer first), so the
        wrapper may hand out its own buffer instead of a fresh clone per step (see VecTaskPythonWrapper.alias_obs)."""
        if hasattr(self.vec_env, "alias_obs"):
            self.vec_env.alias_obs = True

    def env_step(self, actions):
        if self.clip_actions and getattr(self.vec_env, "clip_actions", None) != 1.0:
            # rescale_actions(low, high, clamp(a, -1, 1)) with +-1 spaces is the clamp itself; VecTaskPython.step applies the
            # identical clamp again (vec_task.py:147), so when the wrapper clamps at 1.0 the first one is a no-op and is skipped
            actions = torch.clamp(actions, -1.0, 1.0)
        obs, rewards, dones, infos = self.vec_env.step(actions)
        if self.value_size == 1:
            rewards = rewards.unsqueeze(1)
        return self.obs_to_tensors(obs), rewards, dones, infos

    # ------------------------------------------------------------------ inference (common_agent.py:262-288, 551-562)
    def _obs_store(self, obs):
        """The pitched allocation behind an (N, obs_dim) observation view (zero copy)."""
        return obs

    def _preproc_obs(self, obs_batch, ws, rows, row_idx=None, raw_out=None):
        """running_mean_std(obs) written into the network's GEMM-ready input buffer.  ``raw_out``: the pass also records the raw rows there
        (the rollout's ``obses`` slot; the caller checked _obs_record_fused)."""
        xp = ws.get("xp")                      # a model whose layer 1 runs on the planar GEMM: the normaliser also writes the operand's planes
        self.running_mean_std.forward(obs_batch, row_idx=row_idx, out=ws["x"], out_cols=self.model.in_pitch, planes=xp, raw_out=raw_out)
        if xp is not None:
            ws["xp_fresh"] = True
        return ws["x"]

    def _obs_record_fused(self, obs, ws, slot_rows):
        """May the normaliser pass of a rollout step also record the raw observation into the experience-buffer slot (one launch instead of
        a 15 MB copy + the pass)?  Needs the plain fp32 normaliser on wide 16-byte aligned rows (no operand planes).  Decided once per buffer set."""
        key = (obs.data_ptr(), obs.stride(0), ws["x"].data_ptr(), slot_rows.stride(0))
        if self._obs_fused_key != key:
            self._obs_fused_key = key
            self._obs_fused_ok = (ws.get("xp") is None and obs.dim() == 2 and
                                  K.rms_copy_supported(obs, self.running_mean_std.mean_size, ws["x"], self.model.in_pitch, slot_rows))
        return self._obs_fused_ok

    def get_action_values(self, obs, slot=None, record_obs=False):
        """Actor + critic inference and action sampling for one rollout step.  With ``slot`` (time
        index) the outputs are produced directly inside the experience buffer.  ``record_obs``: the step's observation still has to go into
        experience-buffer slot ``obses[slot]`` (play_steps): done by the normaliser pass when it can, by a copy otherwise."""
        n = self.num_actors
        net = self.model
        ws = net.workspace(n, train=False)
        net.eval()
        raw = None
        if record_obs:
            rows = self.experience_buffer.slot("obses", slot)
            if self._obs_sink_enabled and self._obs_record_fused(obs["obs"], ws, rows):
                raw = rows
            else:
                self.experience_buffer.update_data("obses", slot, obs["obs"])
        self._preproc_obs(obs["obs"], ws, n, raw_out=raw)
        eb = self.experience_buffer
        t = self.horizon_length
        ap = net.a_pitch
        s = 0 if slot is None else slot
        net.forward(ws, n)
        if self.noise_provider is not None:        # parity tests share pre-drawn noise with the CPU oracle
            noise = self.noise_provider(self.epoch_counter, s)
        elif self._rollout_noise is not None:      # play_steps drew the whole rollout's noise in one launch
            noise = self._rollout_noise[s]
        else:
            noise = torch.randn(n, self.actions_num, device=self.ppo_device, generator=self.noise_generator)
        vm = self.value_mean_std
        K.policy_sample(ws["mu"], ws["mu"].stride(0), net.sigma, noise, self.actions_num, n, self.actions_num,
                        eb.phys["actions"], t * ap, eb.phys["neglogpacs"], t, sigmas=eb.phys["sigmas"], sigmas_stride=t * ap,
                        value_raw=ws["val"], value_stride=ws["val"].stride(0), value_mean=vm.running_mean if vm else None,
                        value_var=vm.running_var if vm else None, values=eb.phys["values"], values_stride=t,
                        mus_out=eb.phys["mus"], mus_out_stride=t * ap, mus_out_off=s * ap,
                        actions_off=s * ap, sigmas_off=s * ap, neglogp_off=s, values_off=s)
        td = eb.tensor_dict
        return {"actions": td["actions"][s], "neglogpacs": td["neglogpacs"][s], "values": td["values"][s], "mus": td["mus"][s],
                "sigmas": td["sigmas"][s], "rnn_states": None}

    def _eval_critic_raw(self, obs_dict):
        """Normalise + critic MLP; the (still normalised) value sits in ws['val']."""
        n = self.num_actors
        net = self.model
        ws = net.workspace(n, train=False)
        net.eval()
        self._preproc_obs(obs_dict["obs"], ws, n)
        net.eval_critic(ws, n)
        return ws

    def _eval_critic(self, obs_dict, out=None):
        n = self.num_actors
        ws = self._eval_critic_raw(obs_dict)
        value = torch.empty(n, 1, device=self.ppo_device) if out is None else out
        if self.normalize_value:
            self.value_mean_std.forward(ws["val"], unnorm=True, out=value, out_cols=1)
        else:
            value.copy_(ws["val"])
        return value

    # ------------------------------------------------------------------ rollout (common_agent.py:290-355)
    def play_steps(self):
        self.set_eval()
        eb = self.experience_buffer
        done_mask = None
        if self.noise_provider is None:            # Normal(mu, sigma).sample() of all T steps: one generator launch instead of T
            if self._rollout_noise is None:
                self._rollout_noise = torch.empty(self.horizon_length, self.num_actors, self.actions_num, device=self.ppo_device)
            self._rollout_noise.normal_(generator=self.noise_generator)
        if self._meter_partials is None:
            self._meter_partials = torch.zeros(self.horizon_length, max(1, min(64, self.num_actors // 128)), 4, device=self.ppo_device)
        set_sink = getattr(self.vec_env, "set_obs_sink", None) if self._obs_sink_enabled else None
        for n in range(self.horizon_length):
            self.obs = self._env_reset_masked(done_mask) if done_mask is not None else self.env_reset([])
            res_dict = self.get_action_values(self.obs, slot=n, record_obs=True)         # (+ eb.update_data("obses", n, obs), fused into the normaliser pass)
            for k in self.update_list:
                eb.update_data(k, n, res_dict[k])
            self._before_env_step(n)
            # next_obses[n]: the step kernel writes the row into the slot itself when the env can (set_obs_sink), a copy otherwise
            sink = set_sink is not None and set_sink(eb.slot("next_obses", n))
            self.obs, rewards, self.dones, infos = self.env_step(self._action_for_env(res_dict))
            if not (sink and self.vec_env.obs_sink_written()):
                eb.update_data("next_obses", n, self.obs["obs"])
            self._after_env_step(n, infos)
            # ONE launch for the step's bookkeeping (:318-347): shaped reward / dones / terminate flags into slot n, episode
            # accumulators, the two AverageMeters, and the done mask that drives the next masked reset.  The bootstrap value of the
            # next observation (:394-398) is NOT evaluated here: the critic and the normaliser statistics are frozen for the whole
            # rollout, so critic(next_obs) is computed after the loop for all T*N rows in full-size batches (_bootstrap_values)
            # instead of T under-filled passes over N rows -- same numbers, a third of the rollout's GEMM time back.
            K.rollout_record(rewards=rewards, dones=self.dones, terminate=infos["terminate"], value_raw=None, value_stride=0,
                             value_mean=None, value_var=None, value_eps=0.0,
                             buf_rewards=eb.phys["rewards"][:, n], buf_next_values=None, buf_dones=eb.phys["dones"][:, n],
                             env_stride=self.horizon_length, current_rewards=self.current_rewards, current_lengths=self.current_lengths,
                             meter_rewards=self.game_rewards.state, meter_lengths=self.game_lengths.state, meter_max_size=self.games_to_track,
                             done_mask=self._done_mask, reward_scale=self.rewards_shaper.scale_value, reward_shift=self.rewards_shaper.shift_value,
                             buf_terminate=eb.phys["terminates"][:, n], meter_partials=self._meter_partials[n])
            done_mask = self._done_mask
        # the AverageMeter updates of the T steps, in step order (the meters are only read between epochs): one launch per rollout
        K.rollout_meters(self._meter_partials, self.game_rewards.state, self.game_lengths.state, self.games_to_track)
        self._pending_done_mask = done_mask
        self._bootstrap_values()

        td = eb.tensor_dict
        mb_rewards = self._rollout_rewards(td)
        mb_advs, mb_returns = ops.discount_values(td["dones"], td["values"], mb_rewards, td["next_values"], self.gamma, self.tau,
                                                  return_returns=True)
        batch_dict = eb.get_transformed_list(rlg.swap_and_flatten01, self.tensor_list)
        batch_dict["returns"] = rlg.swap_and_flatten01(mb_returns)
        batch_dict["advs_raw"] = rlg.swap_and_flatten01(mb_advs)
        batch_dict["played_frames"] = self.batch_size
        return batch_dict

    def _bootstrap_values(self):
        """next_values = value_mean_std.unnorm(critic(norm(next_obses))) * (1 - terminated) for every (env, t) of the rollout
        (amp_agent.py:394-398, common_agent.py:551-562).

        The critic and the normaliser statistics are frozen for the whole rollout, and for an env that was NOT reset after step
        t the observation stored at t+1 IS next_obs[t] -- so its bootstrap value is the value already computed at t+1 (the same
        fused-GEMM arithmetic, bit for bit: output elements do not depend on their tile position).  Only the rows that were
        reset (done at t) and the last step need a critic pass of their own: ~7 k of 131 k rows at cfg2 instead of all of them.
        The row list is compacted on the device; its length is the one host read of the rollout."""
        eb, net = self.experience_buffer, self.model
        n, t = self.num_actors, self.horizon_length
        nxt, out, term = eb.flat("next_obses"), eb.flat("next_values"), eb.flat("terminates")
        vals, dones = eb.flat("values"), eb.flat("dones")
        need = dones.view(n, t) != 0
        need[:, t - 1] = True
        # The shortcut holds only for envs whose step() hands back, for envs that were not reset, exactly the observation the next
        # policy step sees (no auto-reset inside step, no per-step observation noise): the env wrapper declares it
        # (``obs_carries_over``) and the first rollout verifies it on the data; otherwise every row gets its own critic pass.
        if self._boot_shortcut is None:
            ok = bool(getattr(self.vec_env, "obs_carries_over", False))
            if ok and t > 1:
                w = self.obs_shape[0]
                cur, nx = eb.flat("obses").view(n, t, -1)[..., :w], nxt.view(n, t, -1)[..., :w]
                ok = bool(((nx[:, :-1] == cur[:, 1:]).all(dim=-1) | need[:, :-1]).all().item())      # one host read, first epoch only
            self._boot_shortcut = ok
        if self._boot_shortcut:
            out.view(n, t)[:, :-1].copy_(vals.view(n, t)[:, 1:])
        else:
            need[:] = True
        rows = torch.nonzero(need.reshape(-1)).reshape(-1)                 # (count,) -- device->host sync on the count, once per epoch
        count = rows.numel()
        chunk = min(self.minibatch_size, max(n, 1024))
        net.eval()
        ws = net.workspace(chunk, train=chunk == self.minibatch_size)     # the update's workspace doubles as the inference one
        if self._boot_idx is None or self._boot_idx.numel() != chunk:
            self._boot_idx = torch.zeros(chunk, dtype=torch.int64, device=self.ppo_device)
            self._boot_val = torch.zeros(chunk, 1, device=self.ppo_device)
        flat_out = out.view(-1)
        for c in range(0, count, chunk):
            m = min(chunk, count - c)
            idx = self._boot_idx
            idx[:m] = rows[c:c + m]
            if m < chunk:
                idx[m:] = rows[c]                                         # padding rows recompute a needed row (result identical)
            self._preproc_obs(nxt, ws, chunk, row_idx=idx)
            net.eval_critic(ws, chunk)
            if self.normalize_value:
                self.value_mean_std.forward(ws["val"], unnorm=True, out=self._boot_val, out_cols=1)
            else:
                self._boot_val.copy_(ws["val"])
            flat_out[idx[:m]] = self._boot_val[:m, 0]
        out.mul_(1.0 - term.unsqueeze(-1).float())                       # next_vals *= (1 - terminated)

    def _action_for_env(self, res_dict):
        return res_dict["actions"]

    def _rollout_rewards(self, td):
        return td["rewards"]

    def _before_env_step(self, n):
        return

    def _after_env_step(self, n, infos):
        return

    def discount_values(self, mb_fdones, mb_values, mb_rewards, mb_next_values):
        return ops.discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values, self.gamma, self.tau)

    # ------------------------------------------------------------------ dataset (common_agent.py:357-398, 589-599)
    def _calc_advs(self, batch_dict):
        returns, values = batch_dict["returns"], batch_dict["values"]
        b = returns.shape[0]
        adv = torch.empty(b, device=self.ppo_device)
        if self.normalize_advantage:
            K.advantage_normalize(returns.reshape(-1), values.reshape(-1), adv, self._adv_partials)
        else:
            torch.sub(returns.reshape(-1), values.reshape(-1), out=adv)
        return adv

    def prepare_dataset(self, batch_dict):
        returns, values = batch_dict["returns"], batch_dict["values"]
        advantages = self._calc_advs(batch_dict)
        if self.normalize_value:
            values = self.value_mean_std(values.reshape(-1, 1))
            returns = self.value_mean_std(returns.reshape(-1, 1))
        eb = self.experience_buffer
        dataset_dict = {
            "old_values": values, "old_logp_actions": batch_dict["neglogpacs"], "advantages": advantages, "returns": returns,
            "actions": batch_dict["actions"], "obs": batch_dict["obses"], "rnn_states": None, "rnn_masks": None,
            "mu": batch_dict["mus"], "sigma": batch_dict["sigmas"],
            # pitched, un-sliced storage for the fused-gather kernels
            "_obs_store": eb.flat("obses"), "_actions_store": eb.flat("actions"), "_mu_store": eb.flat("mus"),
        }
        self.dataset.update_values_dict(dataset_dict)
        return dataset_dict

    # ------------------------------------------------------------------ update (common_agent.py:400-491)
    def train_actor_critic(self, input_dict):
        self.calc_gradients(input_dict)
        return self.train_result

    def _gather_inputs(self, input_dict):
        """Accept the fused form ({'idx', 'dataset'}) or a reference-style gathered dict."""
        if "idx" in input_dict and "dataset" in input_dict:
            d = input_dict["dataset"]
            return (input_dict["idx"], d["_obs_store"], d["_actions_store"], d["_mu_store"], d["old_logp_actions"], d["advantages"],
                    d["old_values"], d["returns"])
        ap = self.model.a_pitch
        def pitched(t, w):
            buf = torch.zeros(t.shape[0], w, device=self.ppo_device)
            buf[:, :t.shape[1]] = t
            return buf
        return (None, pitched(input_dict["obs"], self.obs_pitch), pitched(input_dict["actions"], ap), pitched(input_dict["mu"], ap),
                input_dict["old_logp_actions"].contiguous(), input_dict["advantages"].contiguous(),
                input_dict["old_values"].contiguous(), input_dict["returns"].contiguous())

    def _obs_normalizer_for_update(self):
        """CommonAgent normalises minibatches with the LIVE statistics (and updates them)."""
        return self.running_mean_std, None

    def calc_gradients(self, input_dict):
        self.set_train()
        idx, obs_store, act_store, mu_store, old_nlp, adv, old_val, ret = self._gather_inputs(input_dict)
        mb = idx.numel() if idx is not None else obs_store.shape[0]
        net = self.model
        ws = net.workspace(mb, train=True)
        norm, live = self._obs_normalizer_for_update()
        xp = ws.get("xp")         # layer 1 on the planar GEMM: the normaliser also writes the operand's planes (see _preproc_obs)
        b16 = bool(ws.get("b16"))  # mixed_precision on bf16 storage: the normaliser's output IS the bf16 layer-1 operand
        x_out = ws["x16"] if b16 else ws["x"]
        if live is not None:      # AMPAgent: output from the frozen copy, live statistics still updated (one pass)
            live.forward(obs_store, row_idx=idx, out=x_out, out_cols=net.in_pitch, norm_with=norm, planes=None if b16 else xp)
        else:
            norm.forward(obs_store, row_idx=idx, out=x_out, out_cols=net.in_pitch, planes=None if b16 else xp)
        if b16:
            ws["x16_fresh"] = True
        elif xp is not None:
            ws["xp_fresh"] = True
        overlap = self.multi_gpu and self.overlap_allreduce and hasattr(net, "w_off")
        # single GPU: the slab reduce of every flat gradient also leaves the sums of squares the norm clip needs (with data parallelism the
        # norm is taken after the all-reduce, so it keeps its own pass)
        self._sq_done, self._sq_fuse = set(), not self.multi_gpu
        if self._sq_fuse:
            self._sq_slice(0)                                        # (sizes the shared partials buffer before either chain takes its slice)
        # Independent second chain (AMPAgent: the discriminator's forward / loss / backward shares nothing with the actor / critic chain until
        # the optimiser step): enqueued on a side stream so the two chains' kernels interleave on the chip -- a GEMM launch here is one round
        # of one workgroup per CU, whose prologue, epilogue and launch gap leave CUs idle that the other chain's kernels can use.
        side = self._side_stream()
        extra_info = None
        if side is not None:
            main = torch.cuda.current_stream()
            self._ev_fork.record(main)
            side.wait_event(self._ev_fork)                           # everything enqueued so far (previous optimiser step, the dataset) is visible
            with torch.cuda.stream(side):
                extra_info = self._extra_gradients(input_dict, idx)
                self._ev_join.record(side)
            # what the chain hands back was allocated from the side stream's pool but is read on the main stream (logging, the epoch-end reduce):
            # tell the caching allocator, or a free on one stream could hand the block to the other while it is still in use (round-4 advisor)
            _record_stream(extra_info, main)
        net.forward(ws, mb)
        ap = net.a_pitch
        g16 = {}
        if b16:                   # bf16 copies of d loss / d (mu, value): the operands of the bf16-storage backward
            hp = ws["head_pitch16"]
            g16 = dict(dmu16=ws["dheads16"], dmu16_stride=2 * hp, dvalue16=ws["dheads16"], dvalue16_stride=2 * hp, dvalue16_off=hp)
        K.ppo_loss(mu=ws["mu"], mu_stride=ws["mu"].stride(0), value=ws["val"], value_stride=ws["val"].stride(0), logstd=net.sigma,
                   old_logstd=net.sigma, idx=idx,
                   actions=act_store, actions_stride=act_store.stride(0), old_mu=mu_store, old_mu_stride=mu_store.stride(0),
                   old_neglogp=old_nlp, advantages=adv, old_values=old_val, returns=ret, rows=mb, num_actions=self.actions_num,
                   e_clip=self.e_clip, critic_coef=self.critic_coef, bounds_loss_coef=self.bounds_loss_coef, clip_value=self.clip_value,
                   dmu=ws["dmu"], dmu_stride=ws["dmu"].stride(0), dvalue=ws["dval"], dvalue_stride=ws["dval"].stride(0),
                   partials=self._loss_slot(), **g16)
        bkw = {}
        if self._sq_fuse and getattr(net, "supports_fused_sqnorm", False):
            bkw["sq_partials"] = self._sq_slice(0)
            self._sq_done.add(0)
        if overlap:
            bkw["on_bucket"] = self._bucket_ready
        net.backward(ws, mb, grad_scale=1.0 / self.world_size, **bkw)
        if side is None:
            extra_info = self._extra_gradients(input_dict, idx)           # AMPAgent: discriminator loss / gradients
        else:
            torch.cuda.current_stream().wait_event(self._ev_join)         # the optimiser step needs both chains' gradients
        self._sq_fuse = False
        self._apply_gradients(policy_synced=overlap, sq_done=self._sq_done)
        info, gnorm = self._loss_info(mb)                               # [a_loss, c_loss, b_loss, clip_frac, kl]
        if self._entropy is None:
            ent = float((0.5 + 0.5 * math.log(2 * math.pi)) * self.actions_num) + float(net.sigma.sum().item())
            self._entropy = torch.tensor(ent, device=self.ppo_device)
        self.train_result = {"entropy": self._entropy, "kl": info[4], "last_lr": self.last_lr, "lr_mul": 1.0, "b_loss": info[2],
                             "actor_loss": info[0], "actor_clip_frac": info[3], "critic_loss": info[1],
                             "grad_norm": gnorm}
        self.train_result.update(extra_info)

    def _extra_gradients(self, input_dict, idx):
        return {}

    def _side_stream(self):
        """The stream _extra_gradients runs on beside the actor / critic chain, or None (run it inline).  Only agents whose extra chain
        touches no buffer of the main chain return one."""
        return None

    # Per-minibatch loss statistics.  With a constant LR schedule nothing reads them inside the epoch, so the per-block partial
    # sums of every minibatch are parked in a ring and reduced ONCE at the end of train_epoch (the dicts handed out meanwhile hold
    # views of that epoch's result buffer); an adaptive schedule needs the KL right away and takes the eager path.
    def _begin_loss_ring(self, slots):
        self._lazy_info = not self.is_adaptive_lr
        self._ring_pos, self._ring_mb = 0, []
        if not self._lazy_info:
            return
        p = self._loss_partials.shape
        if self._partials_ring is None or self._partials_ring.shape[0] < slots:
            self._partials_ring = torch.zeros(slots, p[0], p[1], device=self.ppo_device)
        self._info_all = torch.zeros(slots, p[1], device=self.ppo_device)        # fresh per epoch: last epoch's dicts stay valid
        self._gn_all = torch.zeros(slots, device=self.ppo_device)

    def _loss_slot(self):
        if self._lazy_info and self._ring_pos < self._partials_ring.shape[0]:
            return self._partials_ring[self._ring_pos]
        return self._loss_partials

    def _grad_norm_slot(self):
        if self._lazy_info and self._ring_pos < self._gn_all.shape[0]:
            return self._gn_all[self._ring_pos:self._ring_pos + 1]
        return self._grad_norm

    def _loss_info(self, mb):
        if self._lazy_info and self._ring_pos < self._partials_ring.shape[0]:
            i = self._ring_pos
            self._ring_pos += 1
            self._ring_mb.append(mb)
            return self._info_all[i], self._gn_all[i]
        return self._loss_partials.sum(0) / mb, self._grad_norm.clone()

    def _end_loss_ring(self):
        if self._lazy_info and self._ring_pos:
            n = self._ring_pos
            torch.sum(self._partials_ring[:n], dim=1, out=self._info_all[:n])
            if len(set(self._ring_mb)) == 1:
                self._info_all[:n] /= self._ring_mb[0]
            else:
                self._info_all[:n] /= torch.tensor(self._ring_mb, dtype=torch.float32, device=self.ppo_device)[:, None]
        self._lazy_info = False

    def _param_groups(self):
        """(params, grads, exp_avg, exp_avg_sq, count) of every flat buffer the optimiser owns."""
        net = self.model
        return [(net.flat, net.grad, self.exp_avg, self.exp_avg_sq, net.n_flat)]

    SQ_BLOCKS = 1024

    def _sq_slice(self, i):
        """The sum-of-squares partials of parameter group i (one buffer for all groups: the clip takes ONE norm over all of them)."""
        nb, n = self.SQ_BLOCKS, len(self._param_groups())
        if self._sq_partials.numel() != nb * n:
            self._sq_partials = torch.zeros(nb * n, device=self.ppo_device)
        return self._sq_partials[i * nb:(i + 1) * nb]

    def _bucket_ready(self, grad_view):
        """Data-parallel overlap: a finished gradient bucket goes on the wire while the backward continues."""
        self.dist.sync_gradients(grad_view, async_op=True)

    def _apply_gradients(self, policy_synced=False, sq_done=()):
        """[all-reduce] -> clip_grad_norm_ over ALL parameters -> Adam, one fused launch per flat buffer."""
        groups = self._param_groups()
        if self.multi_gpu:
            for i, g in enumerate(groups):
                if i == 0 and policy_synced:
                    continue                                                 # its buckets went out during the backward
                self.dist.sync_gradients(g[1], async_op=policy_synced)       # optimizer.synchronize()
            self.dist.wait_gradients()
        self.optimizer_step += 1
        for i, g in enumerate(groups):
            if i not in sq_done:             # (groups whose gradient reduce already left its sum-of-squares partials)
                K.sqnorm_partial(g[1], g[4], self._sq_slice(i))
        kw = dict(lr=self.last_lr, step=self.optimizer_step, weight_decay=self.weight_decay, max_norm=self.grad_norm if self.truncate_grads else 0.0,
                  sqnorm_partials=self._sq_partials, grad_norm_out=self._grad_norm_slot())
        if 1 < len(groups) <= 4:             # one optimiser over several flat buffers (AMPAgent: policy + discriminator): one launch
            K.adam_step_multi(groups, **kw)
        else:
            for g in groups:
                K.adam_step(g[0], g[1], g[2], g[3], g[4], **kw)

    # ------------------------------------------------------------------ epoch (common_agent.py:191-260)
    def train_epoch(self):
        if not self._tensors_ready:
            self.init_tensors()
            self.obs = self.env_reset()
        torch.cuda.synchronize()
        play_time_start = time.time()
        batch_dict = self.play_steps()
        torch.cuda.synchronize()
        play_time_end = time.time()
        update_time_start = time.time()
        self.set_train()
        self.curr_frames = batch_dict.pop("played_frames")
        self.prepare_dataset(batch_dict)
        train_info = None
        self._begin_loss_ring(self.mini_epochs_num * len(self.dataset))
        for _ in range(0, self.mini_epochs_num):
            for i in range(len(self.dataset)):
                curr_train_info = self.train_actor_critic(self.dataset[i])
                if self.schedule_type == "legacy":
                    if self.multi_gpu and self.is_adaptive_lr:
                        curr_train_info["kl"] = self.dist.average_value(curr_train_info["kl"], "ep_kls")
                    if self.is_adaptive_lr:   # the constant schedule ignores kl: no device read (reference reads kl.item())
                        self.last_lr, self.entropy_coef = self.scheduler.update(self.last_lr, self.entropy_coef, self.epoch_num, 0,
                                                                                curr_train_info["kl"].item())
                        self.update_lr(self.last_lr)
                if train_info is None:
                    train_info = {k: [v] for k, v in curr_train_info.items()}
                else:
                    for k, v in curr_train_info.items():
                        train_info[k].append(v)
            if self.schedule_type == "standard" and self.is_adaptive_lr:
                av_kls = torch.stack(train_info["kl"]).mean()
                if self.multi_gpu:
                    av_kls = self.dist.average_value(av_kls, "ep_kls")
                self.last_lr, self.entropy_coef = self.scheduler.update(self.last_lr, self.entropy_coef, self.epoch_num, 0, av_kls.item())
                self.update_lr(self.last_lr)
        self._end_loss_ring()
        torch.cuda.synchronize()
        update_time_end = time.time()
        self.epoch_counter += 1
        train_info["play_time"] = play_time_end - play_time_start
        train_info["update_time"] = update_time_end - update_time_start
        train_info["total_time"] = update_time_end - play_time_start
        self._record_train_batch_info(batch_dict, train_info)
        return train_info

    def _record_train_batch_info(self, batch_dict, train_info):
        return

    # ------------------------------------------------------------------ train loop (common_agent.py:100-185)
    def train(self, max_epochs=None):
        self.init_tensors()
        self.obs = self.env_reset()
        if self.multi_gpu:
            self.dist.setup_algo(self.model.flat, (self.model.sigma, self.exp_avg, self.exp_avg_sq))
            for g in self._param_groups()[1:]:
                self.dist.setup_algo(g[0], (g[2], g[3]))
        self._init_train()
        total_time = 0.0
        max_epochs = self.max_epochs if max_epochs is None else max_epochs
        while True:
            epoch_num = self.update_epoch()
            train_info = self.train_epoch()
            if self.multi_gpu:
                self.curr_frames = self.dist.sync_stats(self._stat_modules(), self.curr_frames)
            total_time += train_info["total_time"]
            self.frame += self.curr_frames
            if self.rank == 0:
                fps_step = self.curr_frames / train_info["play_time"]
                fps_total = self.curr_frames / train_info["total_time"]
                mean_rewards = self.game_rewards.get_mean()
                print(f"epoch: {epoch_num} frames: {self.frame} fps step: {fps_step:.1f} fps total: {fps_total:.1f} "
                      f"reward: {mean_rewards}", flush=True)
            if self.save_freq > 0 and epoch_num % self.save_freq == 0 and self.rank == 0:      # common_agent.py:160-170
                import os
                os.makedirs(self.nn_dir, exist_ok=True)
                self.save(os.path.join(self.nn_dir, self.config.get("name", "Humanoid")))
            if epoch_num >= max_epochs:
                return self.game_rewards.get_mean(), epoch_num

    def _init_train(self):
        return

    def _stat_modules(self):
        return [self.running_mean_std, self.value_mean_std]

    # ------------------------------------------------------------------ checkpoint surface
    def get_full_state_weights(self):
        state = {"model": self.model.state_dict(), "epoch": self.epoch_num, "frame": self.frame,
                 "optimizer": {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.optimizer_step}}
        if self.normalize_input:
            state["running_mean_std"] = self.running_mean_std.state_dict()
        if self.normalize_value:
            state["reward_mean_std"] = self.value_mean_std.state_dict()
        return state

    def set_full_state_weights(self, weights):
        self.model.load_state_dict(weights["model"])
        self.epoch_num = weights.get("epoch", 0)
        self.frame = weights.get("frame", 0)
        if "optimizer" in weights and "exp_avg" in weights["optimizer"]:
            self.exp_avg.copy_(weights["optimizer"]["exp_avg"])
            self.exp_avg_sq.copy_(weights["optimizer"]["exp_avg_sq"])
            self.optimizer_step = int(weights["optimizer"]["step"])
        if self.normalize_input and "running_mean_std" in weights:
            self.running_mean_std.load_state_dict(weights["running_mean_std"])
        if self.normalize_value and "reward_mean_std" in weights:
            self.value_mean_std.load_state_dict(weights["reward_mean_std"])

    def save(self, fn):
        torch.save(self.get_full_state_weights(), fn + ".pth")

    def restore(self, fn):
        self.set_full_state_weights(torch.load(fn, map_location=self.ppo_device))
