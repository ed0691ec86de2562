"""HumanoidIm task on MI355X: the env side of ``env.step()`` without the physics.

Mirrors the buffer contract and step order of the reference class chain
  Humanoid -> HumanoidAMP -> HumanoidAMPTask -> HumanoidIm
  (phc/env/tasks/humanoid.py:1222-1346, humanoid_amp.py:181-210, humanoid_im.py:662-706,853-919,1119-1192)
and the VecTask return convention (phc/env/tasks/vec_task.py:145-162, vec_task_wrappers.py:45-81):

    step(actions):  pre_physics_step -> _physics_step -> post_physics_step
    post_physics_step:  progress_buf += 1 ; refresh ; _compute_reward ; _compute_reset ;
                        _compute_observations ; extras['terminate'|'reward_raw']

The reference issues three chains of TorchScript calls for reward / reset / observations; here
``post_physics_step`` is ONE launch of ``pulse_im_step`` (pulse_amd/csrc/env_step.hip) that writes
``rew_buf``, ``reward_raw``, ``reset_buf``, ``_terminate_buf`` and ``obs_buf`` in place.  The
per-piece methods (``_compute_reward`` ...) are kept -- same names, same buffers -- and launch the
same kernel with a narrower ``what`` mask, so partial recomputes (``reset(env_ids)``) and tests
can call them exactly like the reference.

Buffers: ``obs_buf`` (N, 934) float32 is a view of an (N, 960) allocation (GEMM-ready pitch, zero
pad); ``rew_buf`` (N,), ``reset_buf`` / ``progress_buf`` / ``_terminate_buf`` (N,) int64.
The simulator is injected (pulse_amd/env/sim.py: Isaac Gym is closed source and OUT OF SCOPE).  The reference motion comes
either from pre-recorded frames (RecordedMotion, what the CPU oracle agent replays) or from the HBM-resident ``MotionLib``
(pulse_amd/env/motion_lib.py; AMASS data is not redistributable, so its clips are synthetic).  With the library the step
kernel advances the episode clock, tests pass_time and blends its t / t+1 reference itself, a reset is one launch
(new start time, flag clears, reference-state init), ``cycle_motion`` restarts finished motions in place and ``fut_tracks``
samples future reference frames -- HumanoidIm.{_compute_task_obs, _compute_reward, _compute_reset, _reset_ref_state_init,
_sample_time} (humanoid_im.py:652-654, 708-919, 920-926, 1119-1192) in three launches per control step.
"""
import os

import torch

from .. import ops
from .. import synthetic as syn
from . import env_keys
from .._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS


class Box:
    """Minimal gym.spaces.Box stand-in (gym is not installed)."""

    def __init__(self, low, high, shape):
        import numpy as np
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=np.float32)
        self.high = np.full(self.shape, high, dtype=np.float32)


class HumanoidIm:
    def __init__(self, cfg, sim, motion_lib, device="cuda:0"):
        env = cfg.get("env", cfg)
        if "env" in env and isinstance(env["env"], dict):                       # legacy layout (phc_*_iccv.yaml): options nested under env:
            env = env["env"]
        if isinstance(cfg.get("robot", None), dict):                            # robot/*.yaml switches (humanoid.py:266-280 reads cfg.robot)
            env = dict({k: v for k, v in cfg["robot"].items() if k in ("has_upright_start", "has_dof_subset", "has_shape_obs", "has_weight_obs")}, **env)
        env_keys.audit(env, type(self).__name__)                               # every key is honoured, inert by contract, or raises by name
        self.cfg = cfg
        self.device = torch.device(device)
        self.sim, self._motion_lib = sim, motion_lib
        self.num_envs = sim.num_envs
        if "num_envs" in env and int(env["num_envs"]) != self.num_envs:
            raise ValueError(f"env.num_envs = {env['num_envs']} but the injected simulator has {self.num_envs} envs")
        self.num_bodies = syn.NUM_BODIES
        self.dt = int(env.get("controlFrequencyInv", 2)) / 60.0                 # base_task.py:92-93: control_freq_inv * sim dt (1 / 60 s, sim/default_sim.yaml)
        self.obs_v = int(env.get("obs_v", 6))
        self.self_obs_v = int(env.get("self_obs_v", 1))
        if self.self_obs_v not in (1, 2, 3) or self.obs_v not in (1, 2, 3, 6, 7, 8, 9):
            raise NotImplementedError("self_obs_v 1 | 2 | 3 with obs_v 1 | 2 | 3 | 6 | 7 | 8 | 9 (4 / 5 need the 10-step / one-hot inputs of "
                                      "humanoid_im.py:467-475, which no shipped config enables)")
        self._has_upright_start = bool(env.get("has_upright_start", True))      # robot/smpl_humanoid.yaml:7
        self.past_track_steps = int(env.get("past_track_steps", 5))            # humanoid.py:330 (self_obs_v 2)
        self.force_sensor_joints = list(env.get("force_sensor_joints", ["L_Ankle", "R_Ankle"]))   # humanoid.py:263 (self_obs_v 3)
        self._fut_tracks = bool(env.get("fut_tracks", False))
        self._num_traj_samples = int(env.get("numTrajSamples", 3)) if self._fut_tracks else 1
        self._traj_sample_timestep = 1.0 / float(env.get("trajSampleTimestepInv", 30))   # humanoid_im.py:80-82
        self._use_motion_lib = hasattr(motion_lib, "query")           # HBM-resident MotionLib vs recorded reference frames
        if self._fut_tracks and not self._use_motion_lib:
            raise NotImplementedError("fut_tracks samples future reference frames: needs the MotionLib reference source")
        if self.obs_v in (2, 8) and self._num_traj_samples != 1:
            raise NotImplementedError("obs_v 2 / 8 take one reference sample (the reference views a column for T > 1, humanoid_im.py:1293,1466)")
        if self.obs_v == 2 and not self._use_motion_lib:
            raise NotImplementedError("obs_v 2 needs the reference dof positions: MotionLib reference source")
        self._local_root_obs = bool(env.get("local_root_obs", True))
        self._root_height_obs = bool(env.get("root_height_obs", True))
        self._full_body_reward = bool(env.get("full_body_reward", True))        # humanoid_im.py:37
        self.power_reward = bool(env.get("power_reward", True))                 # env_im.yaml:23
        self.power_coefficient = float(env.get("power_coefficient", 0.0005))     # humanoid_im.py:92
        self.reward_specs = dict(ops.DEFAULT_REWARD_SPECS)
        self.reward_specs.update(env.get("reward_specs", {}))
        self._enable_early_termination = bool(env.get("enableEarlyTermination", True))
        self.max_episode_length = int(env.get("episode_length", 300))
        self.cycle_motion = bool(env.get("cycle_motion", False))
        # zero_out_far (humanoid.py:311-329; True in the PHC teacher configs phc_kp_pnn_iccv.yaml:36 & co): far envs see their own state
        # as the reference plus a direction to walk in, and are paid for approaching it
        self.zero_out_far = bool(env.get("zero_out_far", False))
        self.zero_out_far_train = bool(env.get("zero_out_far_train", True))
        self.close_distance = float(env.get("close_distance", 0.25))
        self.far_distance = float(env.get("far_distance", 3))
        self._zero_out_far_steps = int(env.get("zero_out_far_steps", 90))
        if self.zero_out_far and self._fut_tracks:
            raise NotImplementedError("zero_out_far with fut_tracks: the reference's blocks subtract an (N T, 3) reference from an (N, 3) root "
                                      "(humanoid_im.py:765,816) and fail for T > 1")
        if self.zero_out_far and self.obs_v not in (6, 7, 8, 9):
            raise NotImplementedError("zero_out_far is defined for obs_v 4 | 5 | 6 | 7 | 8 | 9 (humanoid_im.py:761,812)")
        if self.zero_out_far and self.zero_out_far_train and not self._use_motion_lib:
            raise NotImplementedError("zero_out_far_train moves the reference by a per-env offset: needs the MotionLib reference source")
        # four small options of the reference's step composition [r6]
        self.cycle_motion_xp = bool(env.get("cycle_motion_xp", False))          # humanoid.py:314: a cycled motion restarts up to a metre off (humanoid_im.py:1133-1134)
        self._fut_tracks_dropout = bool(env.get("fut_tracks_dropout", False))   # humanoid_im.py:804-810: a future sample's block is zeroed with p = 0.1
        self.add_obs_noise = bool(env.get("add_obs_noise", False))              # humanoid_im.py:691-692: obs += 0.1 N(0, 1)
        self._res_action = bool(env.get("res_action", False))                   # humanoid_im.py:1096-1101: PD targets = reference pose + scaled action
        self.test = False                                                       # flags.test (run_hydra.py:284): dropout / noise are training-time only
        # occl_training (humanoid.py:323-324): tracked bodies whose reference the task observation does not get to see
        self._occl_training = bool(env.get("occl_training", False))
        self._occl_training_prob = float(env.get("occl_training_prob", 0.1))
        if self._fut_tracks_dropout and self.obs_v not in (6, 8, 9):
            raise NotImplementedError("fut_tracks_dropout acts on obs_v 4 | 5 | 6 | 8 | 9 only (humanoid_im.py:761-810)")
        if self._res_action and not self._use_motion_lib:
            raise NotImplementedError("res_action adds the action to the REFERENCE dof positions: needs the MotionLib reference source")
        if (self._fut_tracks_dropout or self.add_obs_noise) and self.self_obs_v == 2:
            raise NotImplementedError("fut_tracks_dropout / add_obs_noise with self_obs_v 2: not built")
        self._noise_gen = None
        self.strict_eval = bool(env.get("strict_eval", False))                  # humanoid.py:320
        self.im_eval = False                                                    # flags.im_eval (run_hydra.py:291): set by the evaluation loop
        self.auto_pmcp = bool(env.get("auto_pmcp", False))                      # humanoid.py:318-319 -> IMAmpAgent.update_training_data
        self.auto_pmcp_soft = bool(env.get("auto_pmcp_soft", False))
        self.shape_resampling_interval = int(env.get("shape_resampling_interval", 100))     # humanoid.py:283 -> AMPAgent.pre_epoch
        self.getup_schedule = bool(env.get("getup_schedule", False))            # humanoid.py:284 (HumanoidImGetup implements the schedule)
        if self.getup_schedule and not hasattr(self, "update_getup_schedule"):
            raise NotImplementedError("getup_schedule: AMPAgent.pre_epoch calls task.update_getup_schedule (amp_agent.py:569-570), which only the "
                                      "get-up tasks define (humanoid_im_getup.py:67-74): use HumanoidImGetup")
        if env.get("stateInit", "Random") not in ("Random", "Start"):
            raise NotImplementedError(f"stateInit {env.get('stateInit')!r}: Random and Start are built (Default / Hybrid need the simulator's "
                                      "default pose, humanoid_amp.py:447-505)")
        track = env.get("trackBodies", syn.SMPL_BODY_NAMES)
        reset = env.get("reset_bodies", syn.RESET_BODY_NAMES)
        self._track_bodies_id = torch.tensor([syn.SMPL_BODY_NAMES.index(b) for b in track], dtype=torch.int32, device=self.device)
        self._reset_bodies_id = torch.tensor([syn.SMPL_BODY_NAMES.index(b) for b in reset], dtype=torch.int32, device=self.device)
        self._termination_distances = torch.full((self.num_bodies,), float(env.get("terminationDistance", 0.25)), device=self.device)
        self._dof_size = syn.NUM_DOF
        self._pd_action_offset = torch.zeros(self._dof_size, device=self.device)
        self._pd_action_scale = torch.ones(self._dof_size, device=self.device)
        self.clip_obs = float("inf")                                            # parse_task.py:68
        # ---- sizes (humanoid.py:653, humanoid_im.py:457-491)
        lib = ops._lib.load()
        self._force_sensor_width = 6 * len(self.force_sensor_joints) if self.self_obs_v == 3 else 0     # humanoid.py:666-667
        self._hist_steps = self.past_track_steps + 1 if self.self_obs_v == 2 else 1                     # humanoid.py:502-503
        # shape / limb-weight observation rows (robot has_shape_obs / has_weight_obs, humanoid.py:266-268, 657-661: + 11 / + 10 columns)
        self._has_shape_obs = bool(env.get("has_shape_obs", False))
        self._has_limb_weight_obs = bool(env.get("has_weight_obs", False))
        if (self._has_shape_obs or self._has_limb_weight_obs) and self.self_obs_v == 2:
            raise NotImplementedError("the reference raises for shape / limb-weight observations with self_obs_v 2 (humanoid.py:1780-1784)")
        self._self_obs_size = lib.pulse_self_obs_width_ex(self.num_bodies, int(self._root_height_obs), self.self_obs_v, self._hist_steps,
                                                          self._force_sensor_width + 11 * self._has_shape_obs + 10 * self._has_limb_weight_obs)
        jt = self._track_bodies_id.numel()
        self._task_obs_size = lib.pulse_task_obs_width(self.obs_v, jt, self._num_traj_samples)
        self.num_obs = self._self_obs_size + self._task_obs_size
        self.num_actions = self._dof_size
        self.obs_pitch = (self.num_obs + 31) // 32 * 32
        # ---- buffers (base_task.py:98-104)
        n, dev = self.num_envs, self.device
        self._obs_store = torch.zeros(n, self.obs_pitch, device=dev)
        self._im_launch_cache = {}                      # per phase: the filled pulse_im_step argument struct (ops.im_step cache=)
        self._obs_sink, self.obs_sink_written = None, False      # set_obs_sink: a second destination for the next step's observation rows
        self.obs_buf = self._obs_store[:, :self.num_obs]
        self.rew_buf = torch.zeros(n, device=dev)
        self.reward_raw = torch.zeros(n, 5 if self.power_reward else 4, device=dev)
        self.reset_buf = torch.ones(n, dtype=torch.int64, device=dev)
        self.progress_buf = torch.zeros(n, dtype=torch.int64, device=dev)
        self._terminate_buf = torch.zeros(n, dtype=torch.int64, device=dev)
        self._cycle_counter = torch.zeros(n, dtype=torch.int64, device=dev)
        self._point_goal = torch.zeros(n, device=dev)                          # humanoid_im.py:84
        if self._occl_training:
            jt = self._track_bodies_id.numel()
            if jt != self.num_bodies or self._fut_tracks or self.obs_v not in (6, 7, 8, 9):
                raise NotImplementedError("occl_training: the reference's mask update indexes columns 9 .. 23 and its reset indexes the mask by body id "
                                          "(humanoid_im.py:1058, 1183): all 24 bodies tracked, one reference sample, obs_v 4 | 5 | 6 | 7 | 8 | 9")
            self.random_occlu_idx = torch.zeros(n, jt, dtype=torch.bool, device=dev)            # humanoid_im.py:85-86
            self.random_occlu_count = torch.zeros(n, jt, dtype=torch.int64, device=dev)
            self._occl_bits = torch.zeros(n, dtype=torch.int32, device=dev)
            self._occl_weights = (1 << torch.arange(jt, device=dev, dtype=torch.int64))
            self._occl_gen = torch.Generator(device=dev)
            self._occl_gen.manual_seed(int(env.get("occl_seed", 909)))
        self._motion_start_times = torch.zeros(n, device=dev)
        self._motion_start_times_offset = torch.zeros(n, device=dev)
        self._pass_time = torch.zeros(n, dtype=torch.bool, device=dev)
        # self_obs_v 2: the last past_track_steps simulated states + the current one, oldest first (humanoid.py:224-228, 1301-1312)
        self._rb_hist = torch.zeros(n, self._hist_steps, self.num_bodies, 13, device=dev) if self.self_obs_v == 2 else None
        # self_obs_v 3: force-sensor readings (vec_sensor_tensor, humanoid.py:179-187) -- the physics stand-in has none: zeros unless
        # the sim object provides ``force_sensor``
        self._force_sensor = (getattr(sim, "force_sensor", None) if self.self_obs_v == 3 else None)
        if self.self_obs_v == 3 and self._force_sensor is None:
            self._force_sensor = torch.zeros(n, self._force_sensor_width, device=dev)
        # humanoid_shapes (N, 17) = [gender, 10 betas, 6 unused] and humanoid_limb_and_weights (N, 10) are filled while the robot assets are
        # built (humanoid.py:739-878: smpl_sim, out of scope); the sim stand-in may provide them, otherwise a fixed synthetic draw per env
        if self._has_shape_obs or self._has_limb_weight_obs:
            g = torch.Generator().manual_seed(int(env.get("shape_seed", 99)))
            shapes = getattr(sim, "humanoid_shapes", None)
            limbs = getattr(sim, "humanoid_limb_and_weights", None)
            self.humanoid_shapes = (shapes if shapes is not None else torch.cat([torch.randint(0, 2, (n, 1), generator=g).float(),
                                                                                torch.randn(n, 16, generator=g)], dim=1)).to(dev).contiguous()
            self.humanoid_limb_and_weights = (limbs if limbs is not None else torch.rand(n, 10, generator=g) + 0.5).to(dev).contiguous()
        self._motion_len_env = motion_lib._motion_lengths
        if self._use_motion_lib:
            self._init_motion_clock(env)
        self.extras = {}
        self.actions = None
        # attributes the agent reaches for (amp_agent.py:59-63; common_agent.py:54); defaults of humanoid.py:105,296-345
        self.temp_running_mean = bool(env.get("temp_running_mean", True))
        self.kin_lr = float(env.get("kin_lr", 5e-4))
        # fitting (humanoid.py:310; True in env_im_vae.yaml:20 and the PHC teacher configs): AMPAgent loads the normaliser statistics of
        # models[0] and freezes them (amp_agent.py:70-76)
        self.fitting = bool(env.get("fitting", False))
        self.models_path = list(env.get("models", []))
        self.save_kin_info = bool(env.get("save_kin_info", False))
        self.only_kin_loss = bool(env.get("only_kin_loss", False))
        self.distill = bool(env.get("distill", False))
        self.z_type = env.get("z_type", None)
        self.kld_coefficient = float(env.get("kld_coefficient", 0.01))
        self.kld_coefficient_min = float(env.get("kld_coefficient_min", 0.001))
        self.ar1_coefficient = float(env.get("ar1_coefficient", 0.005))
        self.kld_anneal = bool(env.get("kld_anneal", True))
        self.use_ar1_prior = bool(env.get("use_ar1_prior", False))
        self.use_vae_prior = bool(env.get("use_vae_prior", False))
        self.use_vae_prior_regu = bool(env.get("use_vae_prior_regu", False))
        self._task_obs_size_detail = {k: env[k] for k in ("embedding_size", "embedding_norm", "z_type", "use_vae_prior",
                                                          "use_vae_clamped_prior", "vae_var_clamp_max") if k in env}
        self._task_obs_size_detail.setdefault("proj_norm", True)
        # ---- AMP observations (humanoid_amp.py:91-118, 296-314): (N, numAMPObsSteps, W) history, slot 0 = current frame
        self._enable_amp_obs = bool(env.get("enable_amp_obs", False))
        if self._enable_amp_obs:
            self._num_amp_obs_steps = int(env.get("numAMPObsSteps", 10))
            self._amp_root_height_obs = bool(env.get("ampRootHeightObs", env.get("root_height_obs", True)))
            self._key_body_ids = torch.tensor([syn.SMPL_BODY_NAMES.index(b) for b in env.get("key_bodies", ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"])],
                                              dtype=torch.int32, device=dev)
            # dof_subset (humanoid.py:396-421): every joint but L_Hand / R_Hand / L_Toe / R_Toe; applied when the robot config says
            # has_dof_subset (True in robot/smpl_humanoid.yaml:6) -> 19 joints, 196 floats per frame, 1960 per window.  For SMPL the
            # subset tensor always exists, so the in-place zeroing of the toe / hand dofs (humanoid_amp.py:636-639, guarded by
            # ``dof_subset is None``) never runs on this humanoid.
            self._has_dof_subset = bool(env.get("has_dof_subset", True))
            removed = {syn.SMPL_BODY_NAMES.index(b) - 1 for b in ("L_Hand", "R_Hand", "L_Toe", "R_Toe")}
            nj = syn.NUM_DOF // 3
            self._amp_joint_ids = [j for j in range(nj) if j not in removed] if self._has_dof_subset else None
            self._amp_zero_joints = ()
            self._num_amp_obs_per_step = ops.amp_obs_width(len(self._amp_joint_ids) if self._has_dof_subset else nj, self._key_body_ids.numel(),
                                                           self._amp_root_height_obs)
            if self._amp_joint_ids is not None:
                self._amp_joint_ids = torch.tensor(self._amp_joint_ids, dtype=torch.int32, device=dev)
            self._amp_obs_buf = torch.zeros(n, self._num_amp_obs_steps, self._num_amp_obs_per_step, device=dev)
            self._amp_fused = os.environ.get("PULSE_AMP_FUSED", "1") != "0"      # fused history update / history init kernels (0: the op-by-op path)
            self._amp_obs_sink = None
            self._add_amp_input_noise = bool(env.get("add_amp_input_noise", False))        # humanoid_amp.py:135, 281-283: demo windows + 0.01 N(0, 1)
            self._curr_amp_obs_buf = self._amp_obs_buf[:, 0]
            self._hist_amp_obs_buf = self._amp_obs_buf[:, 1:]
            self._amp_obs_space = Box(-float("inf"), float("inf"), (self.get_num_amp_obs(),))
        if self.save_kin_info:        # HumanoidImDistill.kin_dict (humanoid_im_distill.py:73-80, 204-205)
            self.kin_dict = {"gt_action": torch.zeros(n, self.num_actions, device=dev),
                             "progress_buf": torch.zeros(n, dtype=torch.int64, device=dev)}
        self._teacher = None
        self.humanoid_type = "smpl"
        self.has_task = True
        self.viewer = None

    def attach_teacher(self, teacher):
        """HumanoidImDistill (humanoid_im_distill.py:44-63): the frozen policy whose action is the distillation target."""
        self._teacher = teacher

    # ------------------------------------------------------------------ reference motion from the HBM-resident library
    def _init_motion_clock(self, env):
        """humanoid_im.py:130-140, 424-455: every env follows motion ``_sampled_motion_ids[e]`` from ``_motion_start_times[e]``,
        translated by ``_global_offset[e]`` (its env origin)."""
        n, dev, lib = self.num_envs, self.device, self._motion_lib
        self._sampled_motion_ids = torch.arange(n, dtype=torch.int64, device=dev) % lib.num_motions()
        self._motion_len_env = lib.get_motion_length(self._sampled_motion_ids).contiguous()
        # _global_offset is zero after a reset (humanoid_im.py:920-923) and only moves when a motion is cycled in place (:1125-1146)
        self._global_offset = torch.zeros(n, 3, device=dev)
        self._clock_gen = torch.Generator(device=dev)
        self._clock_gen.manual_seed(int(env.get("motion_clock_seed", 2024)))
        self._state_init_random = env.get("stateInit", "Random") != "Start"              # HumanoidAMP.StateInit
        self._ref_bufs = {"demo": {}}
        z = lambda *sh: torch.zeros(*sh, device=dev)
        self._track = {"rb_records": z(n, self.num_bodies, 13), "dof_pos": z(n, self._dof_size), "dof_vel": z(n, self._dof_size)}
        self.sim.track(self._track)
        self._reset_phase = torch.zeros(n, device=dev)

    def _ref_query(self, which, shift, with_records=False, time_steps=1):
        res = self._motion_lib.query(self._sampled_motion_ids, offset=self._global_offset, progress=self.progress_buf, step_shift=shift,
                                     dt=self.dt, start_times=self._motion_start_times, start_offsets=self._motion_start_times_offset,
                                     time_steps=time_steps, traj_dt=self._traj_sample_timestep, out=self._ref_bufs[which],
                                     with_records=with_records)
        res["pos"], res["rot"], res["vel"], res["ang"] = res["rg_pos"], res["rb_rot"], res["body_vel"], res["body_ang_vel"]
        return res

    def _ref_now(self):
        return self._motion_lib.now()

    def _ref_next(self):
        return self._motion_lib.next()

    def _motion_kwargs(self, inc=0):
        """Fused-kernel arguments of the motion-library mode: the episode clock (optionally advanced by ``inc``) and the
        library the kernel blends its reference from; the t+1 reference also lands in the buffers the physics stand-in tracks."""
        clock = {"progress_rw": self.progress_buf, "inc": inc, "dt": self.dt, "start_times": self._motion_start_times,
                 "start_offsets": self._motion_start_times_offset, "motion_len": self._motion_len_env, "cycle_motion": self.cycle_motion,
                 "max_episode_length": self.max_episode_length, "pass_time_out": self._pass_time}
        tr = self._track
        motion = {"lib": self._motion_lib, "ids": self._sampled_motion_ids, "offset": self._global_offset, "traj_dt": self._traj_sample_timestep,
                  "track_rb": tr["rb_records"], "track_dof_pos": tr["dof_pos"], "track_dof_vel": tr["dof_vel"]}
        return clock, motion

    # ------------------------------------------------------------------ sizes / spaces
    def get_obs_size(self):
        return self.num_obs

    def get_self_obs_size(self):
        return self._self_obs_size

    def get_task_obs_size(self):
        return self._task_obs_size

    def get_action_size(self):
        return self.num_actions

    def get_running_mean_size(self):
        return (self.get_obs_size(),)

    def get_num_amp_obs(self):
        return self._num_amp_obs_steps * self._num_amp_obs_per_step

    def set_amp_obs_sink(self, rows):
        """The NEXT post_physics_step writes its finished AMP window into ``rows`` ((N, >= S W) float32, any row pitch) as well, and hands
        that view out as extras['amp_obs'] -- the agent passes the experience-buffer slot of the step, so recording the window costs no copy."""
        self._amp_obs_sink = rows

    def set_obs_sink(self, rows):
        """The NEXT post_physics_step also writes its observation rows into ``rows`` ((N, >= obs pitch) float32, any row pitch: the agent passes
        experience-buffer slot ``next_obses[n]``, so recording the row costs a store inside the step kernel instead of a 15 MB copy launch).
        Returns False -- and does nothing -- when the observation is post-processed after the kernel (fut_tracks_dropout / add_obs_noise at
        training time): the caller copies as before.  ``obs_sink_written`` says whether the last step honoured the sink."""
        self.obs_sink_written = False
        if not self.test and (self._fut_tracks_dropout or self.add_obs_noise):
            self._obs_sink = None
            return False
        self._obs_sink = rows
        return True

    def _update_hist_amp_obs(self):
        """humanoid_amp.py:622-631: shift the history by one slot."""
        self._hist_amp_obs_buf.copy_(self._amp_obs_buf[:, 0:self._num_amp_obs_steps - 1].clone())

    def _compute_amp_observations(self, env_mask=None):
        """humanoid_amp.py:632-667 -> current frame into slot 0 of the history."""
        ops.build_amp_observations_smpl(self.sim.rigid_body_state, self.sim.dof_pos, self.sim.dof_vel, self._key_body_ids,
                                        joint_ids=self._amp_joint_ids, zero_joints=self._amp_zero_joints, local_root_obs=self._local_root_obs,
                                        root_height_obs=self._amp_root_height_obs, out=self._curr_amp_obs_buf, env_mask=env_mask)

    def _init_amp_obs(self, mask):
        """_init_amp_obs (humanoid_amp.py:519-563) for the masked envs.  Slot 0 is the current simulated frame.  With the motion
        library the envs were reference-state initialised, so the history slots hold the AMP frames of the motion at
        start - dt * (k + 1) (_init_amp_obs_ref, :531-563; times before the clip clamp to its first frame); with recorded
        frames there is no motion to look back into and the history repeats the first frame (_init_amp_obs_default, :526-529)."""
        self._compute_amp_observations(env_mask=mask)
        s = self._num_amp_obs_steps - 1
        if self._use_motion_lib and self._amp_fused:
            ops.amp_hist_init(self._motion_lib, self._sampled_motion_ids, self._motion_start_times, self.dt, mask, self._amp_obs_buf, self._key_body_ids,
                              joint_ids=self._amp_joint_ids, local_root_obs=self._local_root_obs, root_height_obs=self._amp_root_height_obs)
            return
        if self._use_motion_lib:
            steps = -self.dt * (torch.arange(0, s, device=self.device) + 1)
            times = (self._motion_start_times.unsqueeze(-1) + steps).view(-1)
            n = self.num_envs
            bufs = self._ref_bufs.setdefault("hist", {})
            res = self._motion_lib.query(self._sampled_motion_ids.repeat_interleave(s), times, out=bufs, fields=("rb_records", "dof_pos", "dof_vel"))
            hist = ops.build_amp_observations_smpl(res["rb_records"], res["dof_pos"], res["dof_vel"], self._key_body_ids, joint_ids=self._amp_joint_ids, zero_joints=(),
                                                   local_root_obs=self._local_root_obs, root_height_obs=self._amp_root_height_obs)
            hist = hist.view(n, s, self._num_amp_obs_per_step)
        else:
            hist = self._curr_amp_obs_buf.unsqueeze(1).expand(-1, s, -1)
        self._hist_amp_obs_buf.copy_(torch.where(mask[:, None, None], hist, self._hist_amp_obs_buf))

    def fetch_amp_obs_demo(self, num_samples):
        """humanoid_amp.py:215-284: AMP observation windows of reference motion (synthetic poses here)."""
        s = self._num_amp_obs_steps
        if self._use_motion_lib:
            lib = self._motion_lib
            ids = lib.sample_motions(num_samples, generator=self._clock_gen)
            # _sample_time (humanoid_amp.py:223-224) resolves to HumanoidIm._sample_time (humanoid_im.py:652-654; HumanoidAMP's own
            # :376-380 does the same for smpl): start times on the 1/30 s grid, no frame blending
            t0 = lib.sample_time_interval(ids, generator=self._clock_gen)
            steps = -self.dt * torch.arange(0, s, device=self.device)                  # build_amp_obs_demo, :256-262
            times = (t0.unsqueeze(-1) + steps).view(-1)
            bufs = self._ref_bufs["demo"] if self._ref_bufs["demo"].get("dof_pos", torch.empty(0)).shape[0] == num_samples * s else {}
            res = lib.query(ids.repeat_interleave(s), times, out=bufs, with_records=True)
            self._ref_bufs["demo"] = res
            rb, dp, dv = res["rb_records"], res["dof_pos"], res["dof_vel"]
        else:
            rb, dp, dv = self._motion_lib.sample_demo_states(num_samples * s)
        out = ops.build_amp_observations_smpl(rb, dp, dv, self._key_body_ids, joint_ids=self._amp_joint_ids, zero_joints=(), local_root_obs=self._local_root_obs,
                                              root_height_obs=self._amp_root_height_obs)
        out = out.view(num_samples, s * self._num_amp_obs_per_step)
        if self._add_amp_input_noise:                      # build_amp_obs_demo's last statement (humanoid_amp.py:281-283)
            self._last_amp_noise = torch.randn(out.shape, device=self.device, generator=self._clock_gen if self._use_motion_lib else None)
            out = out + self._last_amp_noise * 0.01
        return out

    def get_task_obs_size_detail(self):
        return self._task_obs_size_detail

    # ------------------------------------------------------------------ step phases
    def step(self, actions):
        if self.save_kin_info:
            # the distillation target for THIS observation is produced before physics advances
            # (HumanoidImDistill.step, humanoid_im_distill.py:143-231; the teacher itself is out of scope)
            # the frozen PHC teacher (learning/teacher.py) when one is attached; a recorded stand-in otherwise
            self.kin_dict["gt_action"] = self._teacher.forward(self._obs_store) if self._teacher is not None else self.sim.gt_action
            self.kin_dict["progress_buf"] = self.progress_buf.clone()
            self.extras["kin_dict"] = self.kin_dict
        self.pre_physics_step(actions)
        self._physics_step()
        self.post_physics_step()

    def _action_to_pd_targets(self, action):
        if self._res_action:
            # humanoid_im.py:1096-1101: residual action around the reference pose the last _compute_task_obs stored (ref_dof_pos = the t + 1
            # reference's dof positions, :841-848 -- here the buffer the fused step writes for the physics stand-in), clamped to +- pi / 2
            # around the simulated pose
            pd_tar = self._track["dof_pos"] + self._pd_action_scale * action
            pd_lower = self.sim.dof_pos - torch.pi / 2
            pd_upper = self.sim.dof_pos + torch.pi / 2
            return torch.maximum(torch.minimum(pd_tar, pd_upper), pd_lower)
        return torch.addcmul(self._pd_action_offset, self._pd_action_scale, action)   # offset + scale * action, humanoid.py:1392-1394

    def pre_physics_step(self, actions):
        self.actions = actions
        if (self.cycle_motion or (self.zero_out_far and self.zero_out_far_train)) and self._use_motion_lib:
            self._update_cycle_count()                      # (the counter only ever leaves zero in these modes)
        if self._occl_training:
            self._update_occl_training()                    # humanoid_im.py:1114-1115
        self.sim.set_dof_position_target_tensor(self._action_to_pd_targets(actions))

    def _physics_step(self):
        self.sim.simulate_and_refresh()                                         # Isaac Gym: OUT OF SCOPE

    def _update_pass_time(self):
        # humanoid_im.py:1120-1123 (+ :1148 when cycle_motion)
        if self.cycle_motion:
            torch.ge(self.progress_buf, self.max_episode_length - 1, out=self._pass_time)
        else:
            t = self.progress_buf * self.dt + self._motion_start_times + self._motion_start_times_offset
            torch.ge(t, self._motion_len_env, out=self._pass_time)

    def _im_step(self, what, env_ids=None, env_mask=None, ref_next=None, inc=0, obs_copy=None):
        need_now = what & (PULSE_IM_REWARD | PULSE_IM_RESET)
        if self._use_motion_lib:
            clock, motion = self._motion_kwargs(inc)
            ref_now = ref_next = None
        else:
            clock = motion = None
            ref_now = self._ref_now() if need_now else None
            ref_next = (ref_next if ref_next is not None else self._ref_next()) if what & PULSE_IM_TASK_OBS else None
        rb = self.sim.rigid_body_state
        extra = {}
        if self.self_obs_v == 2:
            self._rb_hist[:, -1] = rb                      # slot H-1 = the current state; reward / reset / task obs read it there
            rb = self._rb_hist
        elif self.self_obs_v == 3:
            extra["force_sensor"] = self._force_sensor
        if self.obs_v == 2:
            extra["dof_pos"] = self.sim.dof_pos
        if self._has_shape_obs:
            extra["smpl_params"] = self.humanoid_shapes[:, :-6]          # body_shape_params, humanoid.py:1170
        if self._has_limb_weight_obs:
            extra["limb_weights"] = self.humanoid_limb_and_weights
        rc = self._recovery_counter_for_step()
        if rc is not None:
            extra["recovery_counter"] = rc
        if self.zero_out_far:
            extra["zero_out_far"] = {"point_goal": self._point_goal, "close_distance": self.close_distance, "far_distance": self.far_distance}
        if self._occl_training:
            # the reset only masks occluded bodies in _compute_reset's else-branch (humanoid_im.py:1158 vs :1178-1183)
            extra["occl_bits"], extra["occl_reset"] = self._occl_bits, not (self.zero_out_far and self.zero_out_far_train)
        # the step and the masked-reset launch repeat with the same buffers every control step: their argument structs are cached per phase
        cache = self._im_launch_cache.setdefault((what, inc, env_ids is None, env_mask is None), {})
        return ops.im_step(
            rb, cache=cache, what=what, ref_now=ref_now, ref_next=ref_next, upright=self._has_upright_start,
            enable_early_termination=self._enable_early_termination, self_obs_version=self.self_obs_v, **extra,
            time_steps=self._num_traj_samples, dof_force=self.sim.dof_force, dof_vel=self.sim.dof_vel,
            progress=self.progress_buf, pass_time=self._pass_time, cycle_counter=self._cycle_counter,
            track_ids=self._track_bodies_id, reset_ids=self._reset_bodies_id, term_dist=self._termination_distances,
            reset_use_mean=self.im_eval and not self.strict_eval, full_body_reward=self._full_body_reward, obs_version=self.obs_v,
            local_root_obs=self._local_root_obs, root_height_obs=self._root_height_obs, specs=self.reward_specs,
            power_coef=self.power_coefficient, power_reward=self.power_reward, env_ids=env_ids, env_mask=env_mask,
            obs=self._obs_store, obs_cols=self.obs_pitch, rew=self.rew_buf, rew_raw=self.reward_raw,
            reset=self.reset_buf, terminate=self._terminate_buf, clock=clock, motion=motion, obs_copy=obs_copy)

    def _recovery_counter_for_step(self):
        """HumanoidImGetup hands its recovery counter to the fused step (its _compute_reset, humanoid_im_getup.py:203-210); None here."""
        return None

    def _cycle_motion_update(self):
        """humanoid_im.py:1125-1146 (cycle_motion, neither cycle_motion_xp nor zero_out_far): envs whose motion time ran past the clip
        get a new start time, a clock offset that cancels progress_buf, 60 recovery steps and a global offset that puts the reference
        root under the simulated root.  Sync-free masked form."""
        lib, ids, p = self._motion_lib, self._sampled_motion_ids, self.progress_buf
        t = p * self.dt + self._motion_start_times + self._motion_start_times_offset
        ended = t >= self._motion_len_env
        new_start = lib.sample_time_interval(ids, generator=self._clock_gen)
        self._last_cycle_start = new_start                  # (tests replay the draw on the CPU twin)
        torch.where(ended, -p * self.dt, self._motion_start_times_offset, out=self._motion_start_times_offset)
        torch.where(ended, new_start, self._motion_start_times, out=self._motion_start_times)
        self._cycle_counter.masked_fill_(ended, 60)
        root = lib.get_root_pos_smpl(ids, self._motion_start_times)["root_pos"]
        xy = self.sim.rigid_body_state[:, 0, 0:2] - root[:, 0:2]
        if self.cycle_motion_xp:                                # ... up to a metre away per axis (:1133-1134); drawn for every env (sync-free)
            self._last_xp_uniforms = torch.rand(self.num_envs, 2, device=self.device, generator=self._clock_gen)
            xy = xy + self._last_xp_uniforms
        elif self.zero_out_far and self.zero_out_far_train:     # restart up to 5 m away from the reference (:1135-1142)
            xy = xy + self._far_start_xy()
        self._global_offset[:, 0:2] = torch.where(ended[:, None], xy, self._global_offset[:, 0:2])

    def _far_start_xy(self):
        """humanoid_im.py:937-943 / :1137-1142: a point uniform over the disc of radius max_distance = 5 m, per env (sync-free: drawn for
        every env, the caller keeps the rows it needs).  The draws are kept for the CPU twin (column 0 -> distance, 1 -> angle)."""
        u = torch.rand(self.num_envs, 2, device=self.device, generator=self._clock_gen)
        self._last_far_uniforms = u
        rand_distance = torch.sqrt(u[:, 0]) * 5
        rand_angle = u[:, 1] * torch.pi * 2
        return torch.stack([torch.cos(rand_angle) * rand_distance, torch.sin(rand_angle) * rand_distance], dim=-1)

    def _update_occl_training(self):
        """HumanoidIm._update_occl_training, humanoid_im.py:1046-1058, statement for statement: bodies start an occlusion of 30 .. 59 steps with
        probability occl_training_prob per step (never the root) -- and then the reference OVERWRITES the mask (its last two statements: every
        tracked body occluded except columns 9 .. 23), so what the observation sees is bodies 0 .. 8 hidden; the draws only advance the counters."""
        n, jt = self.random_occlu_idx.shape
        idx = torch.bernoulli(torch.full((n, jt), self._occl_training_prob, device=self.device), generator=self._occl_gen).bool()
        idx[:, 0] = False
        fresh = torch.randint(30, 60, (n, jt), device=self.device, generator=self._occl_gen)       # (drawn for every slot: sync-free masked form)
        self.random_occlu_count = torch.where(idx, fresh, self.random_occlu_count)
        self.random_occlu_count -= 1
        self.random_occlu_count = torch.clamp_min(self.random_occlu_count, 0)
        self.random_occlu_idx = self.random_occlu_count > 0
        self.random_occlu_idx[:] = True
        self.random_occlu_idx[:, [9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23]] = False
        self._pack_occl_bits()

    def _pack_occl_bits(self):
        """random_occlu_idx (N, Jt) bool -> one 32-bit word per env for the fused step (bit j = tracked body j)."""
        self._occl_bits.copy_((self.random_occlu_idx.to(torch.int64) * self._occl_weights).sum(dim=1).to(torch.int32))

    def resample_motions(self):
        """HumanoidIm.resample_motions (humanoid_im.py:350-377), called by AMPAgent.pre_epoch every shape_resampling_interval epochs: the
        reference re-draws which AMASS clips are resident (MotionLib.load_motions under the PMCP sampling weights -- AMASS loading is out of
        scope, so a library that can reload exposes ``load_motions``; the synthetic one keeps every clip resident) and restarts every env."""
        if hasattr(self._motion_lib, "load_motions"):
            self._motion_lib.load_motions()
            if self._use_motion_lib:
                self._motion_len_env = self._motion_lib.get_motion_length(self._sampled_motion_ids).contiguous()
        self.reset()

    def _update_cycle_count(self):
        """humanoid_im.py:1042-1045, called from pre_physics_step (:1112)."""
        self._cycle_counter.sub_(1).clamp_(min=0)

    def _compute_reward(self, actions=None):
        self._im_step(PULSE_IM_REWARD)

    def _compute_reset(self):
        self._update_pass_time()
        self._im_step(PULSE_IM_RESET)

    def _compute_observations(self, env_ids=None, env_mask=None, ref_next=None):
        self._im_step(PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS, env_ids=env_ids, env_mask=env_mask, ref_next=ref_next)
        self._obs_post(env_ids=env_ids, env_mask=env_mask)

    def _obs_post(self, env_ids=None, env_mask=None):
        """What the reference applies to a freshly computed observation at training time (``not flags.test``): fut_tracks_dropout
        (_compute_task_obs, humanoid_im.py:804-810: each future sample's block of the task observation is zeroed with probability 0.1) and
        add_obs_noise (_compute_observations, :691-692: obs + 0.1 N(0, 1) over self AND task observation).  Torch ops on the env's own row
        buffer, only for the envs whose observation was just recomputed; the draws are kept for the CPU twin."""
        if self.test or not (self._fut_tracks_dropout or self.add_obs_noise):
            return
        n, dev = self.num_envs, self.device
        if self._noise_gen is None:
            self._noise_gen = torch.Generator(device=dev)
            self._noise_gen.manual_seed(int(self.cfg.get("env", self.cfg).get("obs_noise_seed", 515)))
        sel = None
        if env_ids is not None:
            sel = torch.zeros(n, dtype=torch.bool, device=dev)
            sel[env_ids] = True
        elif env_mask is not None:
            sel = env_mask
        if self._fut_tracks_dropout:
            t = self._num_traj_samples
            u = torch.rand(n, t, device=dev, generator=self._noise_gen)
            self._last_dropout_uniforms = u
            drop = u < 0.1
            if sel is not None:
                drop = drop & sel[:, None]
            task = self._obs_store[:, self._self_obs_size:self.num_obs].unflatten(1, (t, self._task_obs_size // t))      # a view: (N, T, per sample)
            task.masked_fill_(drop[:, :, None], 0.0)
        if self.add_obs_noise:
            z = torch.randn(n, self.num_obs, device=dev, generator=self._noise_gen)
            self._last_obs_noise = z
            if sel is not None:
                z = z * sel[:, None]
            self.obs_buf.add_(z * 0.1)

    def _update_tensor_history(self):
        """humanoid.py:1308-1312, called before the tensors are refreshed (:1320-1321): the state the LAST observation was computed
        from (slot H-1) moves into the history, the oldest step drops out."""
        h = self._rb_hist
        h[:, :-1] = h[:, 1:].clone()

    def _init_tensor_history(self, mask):
        """humanoid.py:1301-1306 for the masked envs: the history is the reset state repeated."""
        cur = self.sim.rigid_body_state.unsqueeze(1).expand(-1, self._hist_steps, -1, -1)
        torch.where(mask[:, None, None, None], cur, self._rb_hist, out=self._rb_hist)

    def post_physics_step(self):
        if self.self_obs_v == 2:
            self._update_tensor_history()
        # progress += 1, pass_time, reward -> reset -> observations (humanoid.py:1316-1328): one launch.  With the motion library
        # the increment, the time-out test and the reference blend (t and t+1) all happen inside it.
        sink, self._obs_sink = self._obs_sink, None          # set_obs_sink: the observation rows also go to the caller's buffer
        if self._use_motion_lib and self.cycle_motion:
            # cycle_motion (env_im_vae.yaml:55): the reward is taken on the OLD clock, then motions that ran out restart in place
            # (_compute_reset, humanoid_im.py:1125-1146), then reset + observations use the new clock
            self._im_step(PULSE_IM_REWARD, inc=1)
            self._cycle_motion_update()
            self._im_step(PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS, obs_copy=sink)
        elif self._use_motion_lib:
            self._im_step(PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS, inc=1, obs_copy=sink)
        else:
            self.progress_buf += 1
            self._update_pass_time()
            self._im_step(PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS, obs_copy=sink)
        self.obs_sink_written = sink is not None
        self._obs_post()
        self.extras["terminate"] = self._terminate_buf
        self.extras["reward_raw"] = self.reward_raw
        if self._enable_amp_obs:                      # HumanoidAMP.post_physics_step (humanoid_amp.py:194-210)
            if self._amp_fused and self._num_amp_obs_per_step % 4 == 0:
                # history shift + current frame (+ the finished window straight into the caller's row, e.g. the experience-buffer slot
                # the agent registered with set_amp_obs_sink) in ONE launch
                sink = self._amp_obs_sink
                ops.build_amp_observations_smpl(self.sim.rigid_body_state, self.sim.dof_pos, self.sim.dof_vel, self._key_body_ids,
                                                joint_ids=self._amp_joint_ids, zero_joints=self._amp_zero_joints, local_root_obs=self._local_root_obs,
                                                root_height_obs=self._amp_root_height_obs, out=self._curr_amp_obs_buf,
                                                hist_steps=self._num_amp_obs_steps, window_out=sink)
                self._amp_obs_sink = None
                self.extras["amp_obs"] = sink[:, :self.get_num_amp_obs()] if sink is not None else self._amp_obs_buf.view(-1, self.get_num_amp_obs())
            else:
                self._update_hist_amp_obs()
                self._compute_amp_observations()
                self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.get_num_amp_obs())

    # ------------------------------------------------------------------ reset
    def reset(self, env_ids=None):
        """Partial reset by env id (vec_task_wrappers.py:54-56, humanoid.py:526-541): None = every env,
        empty list / tensor = no env."""
        if env_ids is None:
            self.reset_masked(torch.ones(self.num_envs, dtype=torch.bool, device=self.device))
            return
        if not isinstance(env_ids, torch.Tensor):
            if len(env_ids) == 0:
                return
            env_ids = torch.tensor(env_ids, dtype=torch.int64, device=self.device)
        if env_ids.numel() == 0:
            return
        mask = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        mask[env_ids] = True
        self.reset_masked(mask)

    def reset_masked(self, mask):
        """Sync-free form of reset(env_ids): mask is a (N,) bool device tensor.
        _reset_envs (humanoid.py:541-560): state init, buffer clears, observation recompute."""
        if self._use_motion_lib:
            # _reset_envs -> _sample_ref_state (humanoid_im.py:966-986) for the masked envs in ONE launch: new start time
            # (phase * motion length), clock and reset / terminate flags cleared, simulator state := reference state there
            if self._state_init_random:
                self._reset_phase.uniform_(0.0, 1.0, generator=self._clock_gen)           # sample_time_interval's torch.rand (HumanoidIm._sample_time)
            sim = self.sim
            self._motion_lib.query(self._sampled_motion_ids, offset=self._global_offset, dt=self.dt, start_offsets=self._motion_start_times_offset,
                                   out={"rb_records": sim.rigid_body_state, "dof_pos": sim.dof_pos, "dof_vel": sim.dof_vel},
                                   fields=("rb_records", "dof_pos", "dof_vel"),
                                   reset={"mask": mask, "phase": self._reset_phase if self._state_init_random else None, "time_interval": True,
                                          "start_times": self._motion_start_times, "progress": self.progress_buf,
                                          "clear0": self.reset_buf, "clear1": self._terminate_buf, "clear2": self._cycle_counter,
                                          "zero_start_offsets": self._motion_start_times_offset, "zero_global_offset": self._global_offset})
            if hasattr(sim, "on_reset"):
                sim.on_reset(mask)
            if self.zero_out_far and self.zero_out_far_train:
                # _reset_ref_state_init (humanoid_im.py:932-946): the simulated state was initialised from the motion WITHOUT an offset;
                # now the reference moves up to 5 m away and early termination is suspended for zero_out_far_steps steps
                xy = self._far_start_xy()
                self._global_offset[:, 0:2] = torch.where(mask[:, None], xy, self._global_offset[:, 0:2])
                self._cycle_counter.masked_fill_(mask, self._zero_out_far_steps)
            if self.self_obs_v == 2:
                self._init_tensor_history(mask)
            self._compute_observations(env_mask=mask)
            if self._enable_amp_obs:
                self._init_amp_obs(mask)
            return
        keep = ~mask
        self.sim.set_env_states_masked(mask)
        self.progress_buf.mul_(keep)
        ref_next = self._motion_lib.next_after_reset()
        self.reset_buf.mul_(keep)
        self._terminate_buf.mul_(keep)
        if self.self_obs_v == 2:
            self._init_tensor_history(mask)
        self._compute_observations(env_mask=mask, ref_next=ref_next)
        if self._enable_amp_obs:
            self._init_amp_obs(mask)


class VecTaskPythonWrapper:
    """VecTaskPython + wrapper: phc/env/tasks/vec_task.py:145-162, vec_task_wrappers.py:45-81,
    plus RLGPUEnv's get_env_info (phc/run_hydra.py:224-243)."""

    def __init__(self, task, rl_device="cuda:0", clip_observations=float("inf"), clip_actions=1.0):
        self.task = task
        self.env = self                  # agent code reaches vec_env.env.task (amp_agent.py:59)
        self.num_envs = task.num_envs
        self.num_obs = task.num_obs
        self.num_actions = task.num_actions
        self.clip_obs, self.clip_actions = clip_observations, clip_actions
        self.obs_space = Box(-float("inf"), float("inf"), (self.num_obs,))
        self.act_space = Box(-1.0, 1.0, (self.num_actions,))
        self.rl_device = rl_device
        # The reference returns a FRESH tensor every step (torch.clamp(obs_buf, ...), vec_task.py:152-157), so a caller may keep
        # it across steps.  pulse_amd's own agents copy the observation into the experience buffer before the next step and opt
        # into the aliased buffer (saves a 16 MB copy per step at 4096 envs); any other caller gets the reference's semantics.
        self.alias_obs = False
        # for envs that were not reset, the observation step() returns IS what the next policy step sees (no auto-reset inside step, no
        # observation noise): lets the agent reuse values[t + 1] as the bootstrap value of step t (CommonAgent._bootstrap_values)
        self.obs_carries_over = True

    def step(self, actions):
        actions_tensor = torch.clamp(actions, -self.clip_actions, self.clip_actions)
        self.task.step(actions_tensor)
        obs = self.task.obs_buf
        if self.clip_obs != float("inf"):
            obs = torch.clamp(obs, -self.clip_obs, self.clip_obs)
        elif not self.alias_obs:
            obs = obs.clone()
        return obs, self.task.rew_buf, self.task.reset_buf, self.task.extras

    def set_obs_sink(self, rows):
        """Ask the task to write the NEXT step's observation rows into ``rows`` as well (HumanoidIm.set_obs_sink).  Only when what step()
        returns IS the task's buffer (no clamp, aliased hand-out); False = not taken, the caller copies."""
        f = getattr(self.task, "set_obs_sink", None)
        if f is None or self.clip_obs != float("inf") or not self.alias_obs:
            return False
        return bool(f(rows))

    def obs_sink_written(self):
        return bool(getattr(self.task, "obs_sink_written", False))

    def reset(self, env_ids=None):
        self.task.reset(env_ids)
        return self.task.obs_buf

    def reset_masked(self, mask):
        self.task.reset_masked(mask)
        return self.task.obs_buf

    def get_number_of_agents(self):
        return 1

    def get_env_info(self):
        info = {"action_space": self.act_space, "observation_space": self.obs_space, "task_obs_size": self.task.get_task_obs_size()}
        if getattr(self.task, "_enable_amp_obs", False):
            info["amp_observation_space"] = self.task._amp_obs_space
            info["enc_amp_observation_space"] = self.task._amp_obs_space      # RLGPUEnv.get_env_info, run_hydra.py:236-239
        return info

    def fetch_amp_obs_demo(self, num_samples):
        return self.task.fetch_amp_obs_demo(num_samples)
