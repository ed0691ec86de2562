"""Downstream tasks on MI355X: HumanoidSpeed / HumanoidReach / HumanoidStrike and their latent-action (Z) forms.

Mirrors (the tasks a frozen PULSE decoder is trained on, README "pulse_z_task" commands):
  HumanoidAMPTask        phc/env/tasks/humanoid_amp_task.py   observation = [self obs | task obs], _update_task / _reset_task hooks
  HumanoidSpeed(Z)       phc/env/tasks/humanoid_speed.py:22-343
  HumanoidReach(Z)       phc/env/tasks/humanoid_reach.py:19-250
  HumanoidStrike(Z)      phc/env/tasks/humanoid_strike.py:21-380
  Humanoid._compute_reset / compute_humanoid_reset   phc/env/tasks/humanoid.py:1066-1075, 1572-1608
Per control step the env side is TWO launches: the self observation of the fused env-step kernel (pulse_im_step, SELF_OBS only)
and pulse_task_step (task observation at its column offset of the same GEMM-ready row, reward, reset / terminate flags).
The target bookkeeping (_update_task / _reset_task: new target speed / position every tarChangeSteps) is sync-free masked
tensor code -- the reference reads ``nonzero()`` back every step.

Physics is out of scope (Isaac Gym): ``sim`` is any object with the tensor surface of pulse_amd/env/sim.py (rigid_body_state,
dof_force, dof_vel, contact_forces and, for strike, target_states / target_contact_forces).
"""
import math

import torch

from .. import ops
from .. import synthetic as syn
from .._lib import PULSE_IM_SELF_OBS, TASK_OBS, TASK_RESET, TASK_REWARD
from . import env_keys
from .humanoid_z import HumanoidZ


class HumanoidTask:
    TASK = None

    def __init__(self, cfg, sim, device="cuda:0"):
        env = cfg.get("env", cfg)
        env_keys.audit(env, type(self).__name__)             # every key is honoured, inert by contract, or raises by name
        self.cfg, self.sim = cfg, sim
        self.device = torch.device(device)
        self.num_envs, self.num_bodies = sim.num_envs, syn.NUM_BODIES
        self.dt = int(env.get("controlFrequencyInv", 2)) / 60.0             # base_task.py:92-93
        self._local_root_obs = bool(env.get("local_root_obs", True))
        self._root_height_obs = bool(env.get("root_height_obs", True))
        self._has_upright_start = bool(env.get("has_upright_start", True))
        self._enable_task_obs = bool(env.get("enableTaskObs", True))
        self._enable_early_termination = bool(env.get("enableEarlyTermination", True))
        self.max_episode_length = int(env.get("episode_length", 300))
        self.power_reward = bool(env.get("power_reward", False))
        self.power_coefficient = float(env.get("power_coefficient", 0.0005))
        self._dof_size = syn.NUM_DOF
        # power_usage_reward (humanoid_speed.py:43-53, 225-238; humanoid_strike.py:37-40, 186-198): a penalty on the difference between the
        # accumulated |torque x velocity| of the left and the right joints.  Only those two task classes read the key.
        self.power_usage_reward = bool(env.get("power_usage_reward", False)) and self.TASK in ("speed", "strike")
        self.power_usage_coefficient = float(env.get("power_usage_coefficient", 0.0025))
        if self.power_usage_reward:
            dof_names = syn.SMPL_BODY_NAMES[1:]                                        # humanoid.py: _dof_names = body names without the root
            lower = ("Hip", "Knee", "Ankle", "Toe")                                    # the strike task balances the legs only (left_lower_indexes, humanoid.py:425-426)
            pick = lambda side: [i for i, nm in enumerate(dof_names) if nm.startswith(side) and (self.TASK == "speed" or nm[2:] in lower)]
            self._left_idx = torch.tensor(pick("L"), dtype=torch.int64, device=self.device)
            self._right_idx = torch.tensor(pick("R"), dtype=torch.int64, device=self.device)
            self.power_acc = torch.zeros(self.num_envs, 2, device=self.device)
        lib = ops._lib.load()
        self._self_obs_size = lib.pulse_self_obs_width(self.num_bodies, int(self._root_height_obs))
        self._task_obs_size = self._task_obs_width(env) if self._enable_task_obs else 0
        self.num_obs = self._self_obs_size + self._task_obs_size
        self.num_actions = self._dof_size
        self.obs_pitch = (self.num_obs + 31) // 32 * 32
        n, dev = self.num_envs, self.device
        self._obs_store = torch.zeros(n, self.obs_pitch, device=dev)
        self.obs_buf = self._obs_store[:, :self.num_obs]
        self.rew_buf = torch.zeros(n, device=dev)
        self.reward_raw = torch.zeros(n, self._reward_raw_width(), device=dev)
        self.reset_buf = torch.ones(n, dtype=torch.int64, device=dev)
        self.progress_buf = torch.zeros(n, dtype=torch.int64, device=dev)
        self._terminate_buf = torch.zeros(n, dtype=torch.int64, device=dev)
        self._prev_root_pos = torch.zeros(n, 3, device=dev)
        # humanoid.py:237 reads contact_bodies (env_pulse_amp.yaml:63); contactBodies is the legacy spelling (phc_*_iccv.yaml)
        contact = env.get("contact_bodies", env.get("contactBodies", ["R_Ankle", "L_Ankle", "R_Toe", "L_Toe"]))
        self._contact_body_ids = torch.tensor([syn.SMPL_BODY_NAMES.index(b) for b in contact], dtype=torch.int32, device=dev)
        self._termination_heights = torch.full((self.num_bodies,), float(env.get("terminationHeight", 0.15)), device=dev)
        self._task_gen = torch.Generator(device=dev)
        self._task_gen.manual_seed(int(env.get("task_seed", 77)))
        self.extras = {}
        self.humanoid_type, self.has_task, self.viewer = "smpl", True, None
        self.temp_running_mean = bool(env.get("temp_running_mean", True))
        self.save_kin_info = self.only_kin_loss = False
        self.z_type = env.get("z_type", None)

    # ---- sizes
    def _task_obs_width(self, env):
        return ops._lib.load().pulse_task_obs_size({"speed": 1, "reach": 2, "strike": 3}[self.TASK])

    def _reward_raw_width(self):
        return 2 if self.power_reward else 1

    def get_obs_size(self):
        return self.num_obs

    def get_self_obs_size(self):
        return self._self_obs_size

    def get_task_obs_size(self):
        return self._task_obs_size

    def get_running_mean_size(self):
        return (self.num_obs,)

    def get_task_obs_size_detail(self):
        return {}

    # ---- per-task hooks
    def _task_kwargs(self):
        raise NotImplementedError

    def _reset_task(self, mask):
        raise NotImplementedError

    def _task_due(self):
        raise NotImplementedError

    def _update_task(self):
        self._reset_task(self._task_due())                       # masked: envs whose change step has come (no nonzero() read-back)

    def _rand(self, *shape):
        return torch.rand(*shape, device=self.device, generator=self._task_gen)

    def _randint(self, low, high):
        return torch.randint(low, high, (self.num_envs,), device=self.device, dtype=torch.int64, generator=self._task_gen)

    # ---- step phases (humanoid.py:1077-1110, 1315-1331; humanoid_amp_task.py:68-83)
    def _task_step(self, what, env_mask=None):
        kw = self._task_kwargs()
        return ops.task_step(self.TASK, self.sim.rigid_body_state, what=what, prev_root_pos=self._prev_root_pos, dt=self.dt,
                             contact_forces=self.sim.contact_forces, contact_body_ids=self._contact_body_ids,
                             termination_heights=self._termination_heights, progress=self.progress_buf,
                             max_episode_length=float(self.max_episode_length), enable_early_termination=self._enable_early_termination,
                             dof_force=self.sim.dof_force if self.power_reward else None, dof_vel=self.sim.dof_vel if self.power_reward else None,
                             power_coef=self.power_coefficient, power_reward=self.power_reward, obs=self._obs_store, obs_offset=self._self_obs_size,
                             rew=self.rew_buf, rew_raw=self.reward_raw, reset=self.reset_buf, terminate=self._terminate_buf, env_mask=env_mask, **kw)

    def _compute_observations(self, env_mask=None):
        ops.im_step(self.sim.rigid_body_state, what=PULSE_IM_SELF_OBS, local_root_obs=self._local_root_obs, root_height_obs=self._root_height_obs,
                    upright=self._has_upright_start, obs=self._obs_store, obs_cols=self._self_obs_size, env_mask=env_mask)
        if self._enable_task_obs:
            self._task_step(TASK_OBS, env_mask=env_mask)

    def pre_physics_step(self, actions):
        self.actions = actions
        self._prev_root_pos.copy_(self.sim.rigid_body_state[:, 0, 0:3])                 # humanoid_speed.py:72-75
        self.sim.set_dof_position_target_tensor(actions)
        self._update_task()          # HumanoidAMPTask.pre_physics_step (humanoid_amp_task.py:57-59): BEFORE progress_buf advances

    def step(self, actions):
        self.pre_physics_step(actions)
        self.sim.simulate_and_refresh()                                                  # Isaac Gym: OUT OF SCOPE
        self.post_physics_step()

    def post_physics_step(self):
        self.progress_buf += 1
        # reward -> reset (one launch), then the observation row for the next step (humanoid.py:1322-1325)
        self._task_step(TASK_REWARD | TASK_RESET)
        raw = self.reward_raw
        if self.power_usage_reward:
            raw = self._add_power_usage_reward(raw)
        self._compute_observations()
        self.extras["terminate"] = self._terminate_buf
        self.extras["reward_raw"] = raw

    def _add_power_usage_reward(self, raw):
        """humanoid_speed.py:225-238 / humanoid_strike.py:186-198, statement for statement on the kernel's reward (torch ops, as in the
        reference: the option is off in every shipped config).  The speed task appends the term to reward_raw, the strike task does not."""
        power_all = torch.abs(torch.multiply(self.sim.dof_force, self.sim.dof_vel)).reshape(-1, self._dof_size // 3, 3)
        left_power = power_all[:, self._left_idx].reshape(self.num_envs, -1).sum(dim=-1)
        right_power = power_all[:, self._right_idx].reshape(self.num_envs, -1).sum(dim=-1)
        self.power_acc[:, 0] += left_power
        self.power_acc[:, 1] += right_power
        pur = self.power_acc / (self.progress_buf + 1)[:, None]
        pur = -self.power_usage_coefficient * (pur[:, 0] - pur[:, 1]).abs()
        pur = torch.where(self.progress_buf <= 3, torch.zeros_like(pur), pur)           # (pur[progress <= 3] = 0 without a host read)
        self.rew_buf += pur
        return torch.cat([raw, pur[:, None]], dim=-1) if self.TASK == "speed" else raw

    def reset(self, env_ids=None):
        if env_ids is None:
            mask = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        else:
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
        """_reset_envs (humanoid.py:574-587) + HumanoidAMPTask._reset_envs (task reset before the observation)."""
        keep = ~mask
        self.sim.set_env_states_masked(mask)
        self.progress_buf.mul_(keep)
        self.reset_buf.mul_(keep)
        self._terminate_buf.mul_(keep)
        if self.power_usage_reward and self.TASK == "speed":
            self.power_acc.masked_fill_(mask[:, None], 0.0)     # HumanoidSpeed._reset_ref_state_init (humanoid_speed.py:247-249); the strike task never clears it
        self._reset_task(mask)
        self._compute_observations(env_mask=mask)


class HumanoidSpeed(HumanoidTask):
    TASK = "speed"

    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        env = cfg.get("env", cfg)
        self._tar_speed_min, self._tar_speed_max = float(env.get("tarSpeedMin", 0.0)), float(env.get("tarSpeedMax", 5.0))
        self._speed_change_steps_min = int(env.get("speedChangeStepsMin", 100))
        self._speed_change_steps_max = int(env.get("speedChangeStepsMax", 200))
        self._speed_change_steps = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)
        self._tar_speed = torch.ones(self.num_envs, device=self.device)

    def _task_due(self):
        return self.progress_buf >= self._speed_change_steps                               # humanoid_speed.py:153-160

    def _reset_task(self, mask):
        """humanoid_speed.py:162-171 for the masked envs."""
        tar = (self._tar_speed_max - self._tar_speed_min) * self._rand(self.num_envs) + self._tar_speed_min
        steps = self._randint(self._speed_change_steps_min, self._speed_change_steps_max)
        torch.where(mask, tar, self._tar_speed, out=self._tar_speed)
        torch.where(mask, self.progress_buf + steps, self._speed_change_steps, out=self._speed_change_steps)

    def _task_kwargs(self):
        return {"tar_speed": self._tar_speed}


class HumanoidReach(HumanoidTask):
    TASK = "reach"

    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        env = cfg.get("env", cfg)
        self._tar_change_steps_min, self._tar_change_steps_max = int(env.get("tarChangeStepsMin", 100)), int(env.get("tarChangeStepsMax", 200))
        self._tar_dist_max = float(env.get("tarDistMax", 1.0))
        self._tar_height_min, self._tar_height_max = float(env.get("tarHeightMin", 0.6)), float(env.get("tarHeightMax", 1.6))
        self._reach_body_id = syn.SMPL_BODY_NAMES.index(env.get("reachBodyName", "R_Hand"))
        self._tar_change_steps = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)
        self._tar_pos = torch.zeros(self.num_envs, 3, device=self.device)

    def _task_due(self):
        return self.progress_buf >= self._tar_change_steps                                # humanoid_reach.py:122-127

    def _reset_task(self, mask):
        """humanoid_reach.py:129-142 for the masked envs."""
        r = self._rand(self.num_envs, 3)
        r[:, 0:2] = self._tar_dist_max * (2.0 * r[:, 0:2] - 1.0)
        r[:, 2] = (self._tar_height_max - self._tar_height_min) * r[:, 2] + self._tar_height_min
        steps = self._randint(self._tar_change_steps_min, self._tar_change_steps_max)
        torch.where(mask[:, None], r, self._tar_pos, out=self._tar_pos)
        torch.where(mask, self.progress_buf + steps, self._tar_change_steps, out=self._tar_change_steps)

    def _task_kwargs(self):
        return {"tar_pos": self._tar_pos, "reach_body_id": self._reach_body_id}


class HumanoidStrike(HumanoidTask):
    TASK = "strike"

    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        env = cfg.get("env", cfg)
        self._tar_dist_min, self._tar_dist_max = float(env.get("tarDistMin", 0.5)), float(env.get("tarDistMax", 10.0))
        self._near_dist, self._near_prob = float(env.get("nearDist", 1.5)), float(env.get("nearProb", 0.5))
        strike = env.get("strikeBodyNames", ["R_Hand", "R_Wrist", "R_Elbow"])
        self._strike_body_ids = torch.tensor([syn.SMPL_BODY_NAMES.index(b) for b in strike], dtype=torch.int32, device=self.device)
        self._target_states = sim.target_states                                           # (N, 13) root state of the target object

    def _task_due(self):
        return torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)           # the target only moves at episode reset

    def _reset_task(self, mask):
        """_reset_target, humanoid_strike.py:125-145 for the masked envs."""
        n = self.num_envs
        near = self._rand(n) < self._near_prob
        dist_max = torch.where(near, torch.full((n,), self._near_dist, device=self.device), torch.full((n,), self._tar_dist_max, device=self.device))
        dist = (dist_max - self._tar_dist_min) * self._rand(n) + self._tar_dist_min
        theta = 2 * math.pi * self._rand(n)
        rot_theta = 2 * math.pi * self._rand(n)
        root = self.sim.rigid_body_state[:, 0]
        new = torch.zeros(n, 13, device=self.device)
        new[:, 0] = dist * torch.cos(theta) + root[:, 0]
        new[:, 1] = dist * torch.sin(theta) + root[:, 1]
        new[:, 2] = 0.9
        new[:, 5], new[:, 6] = torch.sin(rot_theta / 2), torch.cos(rot_theta / 2)            # quat_from_angle_axis(theta, z)
        torch.where(mask[:, None], new, self._target_states, out=self._target_states)

    def _task_kwargs(self):
        return {"tar_states": self._target_states, "tar_contact_forces": self.sim.target_contact_forces, "strike_body_ids": self._strike_body_ids}


class HumanoidTraj(HumanoidTask):
    """HumanoidTraj (phc/env/tasks/humanoid_traj.py:22-211): follow a random 2-D trajectory (TrajGenerator, phc/utils/traj_generator.py);
    task observation = numTrajSamples (10) future trajectory points in the heading frame, reward exp(-2 |root - target|^2), reset when
    fallen or more than fail_dist (4 m) off the trajectory.  One launch of pulse_traj_step per phase; trajectories are regenerated for the
    envs being reset by pulse_traj_generate from six uniform draws taken in the reference's order."""
    TASK = "traj"
    TERRAIN_OBS = False

    def __init__(self, cfg, sim, device="cuda:0"):
        env = cfg.get("env", cfg)
        self._num_traj_samples = int(env.get("numTrajSamples", 10))
        self._traj_sample_timestep = float(env.get("trajSampleTimestep", 0.5))
        self.terrain_obs = bool(env.get("terrain_obs", self.TERRAIN_OBS))
        self._height_points = self._center_points = self._heightsamples = None
        if self.terrain_obs:
            if env.get("terrain_obs_type", "square") != "square":
                raise NotImplementedError("terrain_obs_type 'square' (env_pulse_terrain.yaml) is built; 'fov' / 'square_fov' are viewer-era variants")
            self._height_points = syn.square_height_points(float(env.get("sensor_extent", 2)), int(env.get("sensor_res", 32))).to(device)
        super().__init__(cfg, sim, device)
        dev = self.device
        self._speed_min, self._speed_max = float(env.get("speedMin", 0.0)), float(env.get("speedMax", 3.0))
        self._accel_max, self._sharp_turn_prob = float(env.get("accelMax", 2.0)), float(env.get("sharpTurnProb", 0.02))
        self._fail_dist = 4.0                                                            # humanoid_traj.py:30
        self._num_verts, self._dtheta_max = 101, 2.0                                     # _build_traj_generator (:105-113)
        self._episode_dur = self.max_episode_length * self.dt
        self._traj_verts = torch.zeros(self.num_envs, self._num_verts, 3, device=dev)
        self.fuzzy_target = bool(env.get("fuzzy_target", False))
        self._sensor_body = 0
        if self.terrain_obs:
            self._center_points = syn.center_height_points().to(dev)
            self._use_center_height = bool(env.get("use_center_height", False))
            self._sensor_body = syn.SMPL_BODY_NAMES.index("Head") if env.get("terrain_obs_root", "head") == "head" else 0
            terrain_type = env.get("terrain", {}).get("terrainType", "trimesh")
            if terrain_type == "none":
                raise NameError("Can't measure height with terrain type 'none'")            # get_heights (:744-745)
            if terrain_type != "plane":
                hs = getattr(sim, "heightsamples", None)
                self._heightsamples = (hs if hs is not None else syn.synthetic_height_field()).to(dev).contiguous()
            self._horizontal_scale, self._vertical_scale, self.height_meas_scale = 0.1, 0.005, 5.0    # Terrain.__init__ (:1121-1122), :69

    def _task_obs_width(self, env):
        w = 2 * self._num_traj_samples                                                   # get_task_obs_size (:253-270)
        if self._height_points is not None:
            w += self._height_points.shape[0]
        return w

    def _reward_raw_width(self):
        return 2                                                                         # [location reward, power reward] (:890)

    def get_task_obs_size_detail(self):
        d = {"traj": 2 * self._num_traj_samples}                                         # get_task_obs_size_detail (:273-290)
        if self.terrain_obs:
            d["heightmap"] = self._height_points.shape[0]
        return d

    def _task_due(self):
        return None

    def _update_task(self):
        return                                                                           # trajectories only change at episode reset

    def _reset_task(self, mask):
        """HumanoidTraj._reset_task (:145-150) -> TrajGenerator.reset for the masked envs."""
        # This runs on EVERY rollout step (the done mask lives on the device): all draws of a call come from ONE uniform launch into a persistent
        # buffer (four (N, V - 1) blocks + two (N,) rows), the sharp-turn Bernoulli is a compare of its block (five RNG launches and
        # six allocations per step before).  The reference draws len(env_ids) rows per reset: the RNG streams were never comparable.
        n, v = self.num_envs, self._num_verts
        blk = n * (v - 1)
        buf = getattr(self, "_traj_draws", None)
        if buf is None or buf.numel() != 4 * blk + 2 * n:
            buf = self._traj_draws = torch.empty(4 * blk + 2 * n, device=self.device)
            self._traj_sharp = torch.empty(n, v - 1, dtype=torch.bool, device=self.device)
        buf.uniform_(generator=self._task_gen)
        u_dtheta, u_sharp, u_dspeed, u_turn = (buf[i * blk:(i + 1) * blk].view(n, v - 1) for i in range(4))
        u_heading, u_speed0 = buf[4 * blk:4 * blk + n], buf[4 * blk + n:]
        sharp_mask = torch.lt(u_turn, self._sharp_turn_prob, out=self._traj_sharp)
        ops.traj_generate(self.sim.rigid_body_state, self._traj_verts, u_dtheta, u_sharp, sharp_mask, u_heading, u_dspeed, u_speed0,
                          episode_dur=self._episode_dur, dtheta_max=self._dtheta_max, speed_min=self._speed_min, speed_max=self._speed_max,
                          accel_max=self._accel_max, env_mask=mask)

    def _task_step(self, what, env_mask=None):
        terrain = {}
        if self.terrain_obs:
            terrain = dict(heightsamples=self._heightsamples, horizontal_scale=self._horizontal_scale, vertical_scale=self._vertical_scale,
                           height_points=self._height_points, sensor_body=self._sensor_body, center_points=self._center_points,
                           use_center_height=self._use_center_height, height_meas_scale=self.height_meas_scale)
        return ops.traj_step(self.sim.rigid_body_state, self._traj_verts, self.progress_buf, what=what, dt=self.dt, episode_dur=self._episode_dur,
                             num_samples=self._num_traj_samples, sample_timestep=self._traj_sample_timestep, upright=self._has_upright_start,
                             dof_force=self.sim.dof_force, dof_vel=self.sim.dof_vel, power_coef=self.power_coefficient, power_reward=self.power_reward,
                             fuzzy_target=self.fuzzy_target, contact_forces=self.sim.contact_forces, contact_body_ids=self._contact_body_ids,
                             termination_heights=self._termination_heights, max_episode_length=float(self.max_episode_length),
                             fail_dist=self._fail_dist, enable_early_termination=self._enable_early_termination, terrain_reset=self.terrain_obs,
                             obs=self._obs_store, obs_offset=self._self_obs_size, rew=self.rew_buf, rew_raw=self.reward_raw, reset=self.reset_buf,
                             terminate=self._terminate_buf, env_mask=env_mask, **terrain)


class HumanoidPedestrianTerrain(HumanoidTraj):
    """HumanoidPedestrianTerrain (phc/env/tasks/humanoid_pedestrian_terrain.py:31-890; env_pulse_terrain.yaml): the trajectory task over a height
    field -- the task observation gains the 32 x 32 height map under the head (:384-440), a fall is a summed non-foot contact force above
    50 N (:1476-1531), the reward keeps a power term (:871-890).  The terrain itself (isaacgym.terrain_utils, PhysX tri-mesh) is a synthetic
    height field here; crowds (``_divide_group``, the 'people' point-net input of amp_sept) are not built -- no shipped config enables them."""
    TERRAIN_OBS = True


class _ZMixin(HumanoidZ):
    """HumanoidSpeedZ / ReachZ / StrikeZ (humanoid_speed.py:290-304 ...): the action is the 32-d latent of a frozen PULSE decoder."""

    def _init_z(self, cfg):
        env = cfg.get("env", cfg)
        self._embedding_size = int(env.get("embedding_size", 32))
        self.num_actions = self._embedding_size

    def step(self, action_z):
        self.action_z = action_z
        actions = self.compute_z_actions(action_z)
        self.pre_physics_step(actions)
        self.sim.simulate_and_refresh()
        self.post_physics_step()


class HumanoidSpeedZ(_ZMixin, HumanoidSpeed):
    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        self._init_z(cfg)


class HumanoidReachZ(_ZMixin, HumanoidReach):
    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        self._init_z(cfg)


class HumanoidStrikeZ(_ZMixin, HumanoidStrike):
    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        self._init_z(cfg)


class HumanoidPedestrianTerrainZ(_ZMixin, HumanoidPedestrianTerrain):
    """humanoid_pedestrian_terrain.py:957-972: the terrain task driven through a frozen PULSE decoder (learning=pulse_z_terrain)."""

    def __init__(self, cfg, sim, device="cuda:0"):
        super().__init__(cfg, sim, device)
        self._init_z(cfg)


class SyntheticTaskSim:
    """Physics stand-in for the downstream tasks: a bank of recorded frames (rigid bodies with a drifting root, sparse contact
    forces, a target object) replayed one per control step.  Actions are accepted and ignored, exactly like RecordedSim."""

    def __init__(self, num_envs, frames, device, seed=1234, rank=0, fall_rate=0.01, xy_offset=(0.0, 0.0)):
        g = syn.make_generator(seed + 31, rank)
        n, j, f = num_envs, syn.NUM_BODIES, frames
        self.num_envs, self.frames, self.frame = n, f, 0
        rb = torch.stack([syn.rigid_body_state(g, n) for _ in range(f)])
        drift = torch.cumsum(0.05 * torch.randn(f, n, 1, 3, generator=g), dim=0)
        drift[..., 2] = 0
        rb[..., 0:3] += drift
        rb[..., 0] += xy_offset[0]                                                             # e.g. into the interior of a height field
        rb[..., 1] += xy_offset[1]
        contact = torch.zeros(f, n, j, 3)
        fell = torch.rand(f, n, generator=g) < fall_rate
        contact[..., 9, 2] = fell.float() * 200.0                                          # torso contact
        rb[..., 9, 2] = torch.where(fell, torch.full_like(rb[..., 9, 2], 0.05), rb[..., 9, 2])
        self.bank = {"rb": rb.to(device), "contact": contact.to(device), "dof_force": (50.0 * torch.randn(f, n, syn.NUM_DOF, generator=g)).to(device),
                     "dof_vel": torch.randn(f, n, syn.NUM_DOF, generator=g).to(device),
                     "tar_contact": (30.0 * torch.randn(f, n, 3, generator=g)).to(device)}
        self.rigid_body_state = self.bank["rb"][0].clone()
        self.contact_forces = self.bank["contact"][0].clone()
        self.dof_force, self.dof_vel = self.bank["dof_force"][0].clone(), self.bank["dof_vel"][0].clone()
        self.dof_pos = torch.zeros(n, syn.NUM_DOF, device=device)
        self.target_states = torch.zeros(n, 13, device=device)
        self.target_states[:, 6] = 1.0
        self.target_contact_forces = self.bank["tar_contact"][0].clone()
        self.pd_targets = None

    def set_dof_position_target_tensor(self, pd_tar):
        self.pd_targets = pd_tar

    def simulate_and_refresh(self):
        self.frame = (self.frame + 1) % self.frames
        f, b = self.frame, self.bank
        self.rigid_body_state.copy_(b["rb"][f])
        self.contact_forces.copy_(b["contact"][f])
        self.dof_force.copy_(b["dof_force"][f])
        self.dof_vel.copy_(b["dof_vel"][f])
        self.target_contact_forces.copy_(b["tar_contact"][f])

    def set_env_states_masked(self, mask):
        return


TASKS = {"HumanoidSpeed": HumanoidSpeed, "HumanoidReach": HumanoidReach, "HumanoidStrike": HumanoidStrike,
         "HumanoidSpeedZ": HumanoidSpeedZ, "HumanoidReachZ": HumanoidReachZ, "HumanoidStrikeZ": HumanoidStrikeZ,
         "HumanoidTraj": HumanoidTraj, "HumanoidPedestrianTerrain": HumanoidPedestrianTerrain,
         "HumanoidPedestrianTerrainZ": HumanoidPedestrianTerrainZ}
