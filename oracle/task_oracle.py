"""CPU restatement of the downstream-task observation / reward / reset functions (ORACLE / TEST INFRASTRUCTURE).

SURVEY.md 8(f) rank 4 -- the tasks of the README's PULSE commands (HumanoidSpeedZ / ReachZ / StrikeZ): oracle first, pinned by
tests/golden/tasks.npz (generated from the reference's own TorchScript functions, oracle/gen_golden.py: gen_tasks); the HIP
kernels for them are the next round's work.  xyzw quaternions throughout.

  humanoid_reset          compute_humanoid_reset              phc/env/tasks/humanoid.py:1572-1608
  speed_observations      compute_speed_observations          phc/env/tasks/humanoid_speed.py:310-324
  speed_reward            compute_speed_reward                phc/env/tasks/humanoid_speed.py:326-343
  location_observations   compute_location_observations       phc/env/tasks/humanoid_reach.py:224-236
  reach_reward            compute_reach_reward                phc/env/tasks/humanoid_reach.py:238-250
  strike_observations     compute_strike_observations         phc/env/tasks/humanoid_strike.py:270-293
  strike_reward           compute_strike_reward               phc/env/tasks/humanoid_strike.py:295-327
  strike_reset            compute_humanoid_reset (strike)     phc/env/tasks/humanoid_strike.py:330-380
"""
import torch

from . import rotations as R


def _fall(contact_buf, contact_body_ids, rigid_body_pos, termination_heights):
    masked = contact_buf.clone()
    masked[:, contact_body_ids, :] = 0
    fall_contact = torch.any(torch.any(torch.abs(masked) > 0.1, dim=-1), dim=-1)
    fall_height = rigid_body_pos[..., 2] < termination_heights
    fall_height[:, contact_body_ids] = False
    return torch.logical_and(fall_contact, torch.any(fall_height, dim=-1)), masked


def humanoid_reset(reset_buf, progress_buf, contact_buf, contact_body_ids, rigid_body_pos, max_episode_length, enable_early_termination,
                   termination_heights):
    terminated = torch.zeros_like(reset_buf)
    if enable_early_termination:
        has_fallen, _ = _fall(contact_buf, contact_body_ids, rigid_body_pos, termination_heights)
        has_fallen = has_fallen * (progress_buf > 1)
        terminated = torch.where(has_fallen, torch.ones_like(reset_buf), terminated)
    reset = torch.where(progress_buf >= max_episode_length - 1, torch.ones_like(reset_buf), terminated)
    return reset, terminated


def speed_observations(root_states, tar_speed):
    tar_dir3d = torch.zeros_like(root_states[..., 0:3])
    tar_dir3d[..., 0] = 1
    local = R.qrot(R.heading_q_inv(root_states[:, 3:7]), tar_dir3d)
    return torch.cat([local[..., 0:2], tar_speed.unsqueeze(-1)], dim=-1)


def speed_reward(root_pos, prev_root_pos, root_rot, tar_speed, dt):
    root_vel = (root_pos - prev_root_pos) / dt
    tar_vel_err = tar_speed - root_vel[..., 0]
    tangent = root_vel[..., 1]
    return torch.exp(-0.25 * (tar_vel_err * tar_vel_err + 0.1 * tangent * tangent))


def power_usage_reward(dof_force, dof_vel, power_acc, progress_buf, left_idx, right_idx, coef):
    """The power_usage_reward block of HumanoidSpeed._compute_reward (humanoid_speed.py:225-238; left_indexes / right_indexes) and
    HumanoidStrike._compute_reward (humanoid_strike.py:186-198; left_lower_indexes / right_lower_indexes): accumulates the per-side
    |torque x velocity| into ``power_acc`` (N, 2) IN PLACE and returns the reward term (N,)."""
    power_all = torch.abs(torch.multiply(dof_force, dof_vel)).reshape(-1, 23, 3)
    n = power_all.shape[0]
    power_acc[:, 0] += power_all[:, left_idx].reshape(n, -1).sum(dim=-1)
    power_acc[:, 1] += power_all[:, right_idx].reshape(n, -1).sum(dim=-1)
    pur = power_acc / (progress_buf + 1)[:, None]
    pur = -coef * (pur[:, 0] - pur[:, 1]).abs()
    pur[progress_buf <= 3] = 0
    return pur


def side_dof_indexes(dof_names, lower_only):
    """humanoid.py:422-426: left_indexes / right_indexes (lower_only: the *_lower_indexes: hips, knees, ankles, toes)."""
    lower = ("Hip", "Knee", "Ankle", "Toe")
    pick = lambda side: [i for i, nm in enumerate(dof_names) if nm.startswith(side) and (not lower_only or nm[2:] in lower)]
    return pick("L"), pick("R")


def location_observations(root_states, tar_pos):
    return R.qrot(R.heading_q_inv(root_states[:, 3:7]), tar_pos - root_states[:, 0:3])


def reach_reward(reach_body_pos, root_rot, tar_pos, tar_speed, dt):
    pos_diff = tar_pos - reach_body_pos
    return torch.exp(-4.0 * torch.sum(pos_diff * pos_diff, dim=-1))


def strike_observations(root_states, tar_states):
    root_pos, root_rot = root_states[:, 0:3], root_states[:, 3:7]
    tar_pos, tar_rot, tar_vel, tar_ang = tar_states[:, 0:3], tar_states[:, 3:7], tar_states[:, 7:10], tar_states[:, 10:13]
    h = R.heading_q_inv(root_rot)
    local_pos = tar_pos - root_pos
    local_pos[..., -1] = tar_pos[..., -1]
    return torch.cat([R.qrot(h, local_pos), R.q_to_tan_norm(R.qmul(h, tar_rot)), R.qrot(h, tar_vel), R.qrot(h, tar_ang)], dim=-1)


def _quat_rotate_isaac(q, v):
    """isaacgym.torch_utils.quat_rotate (what humanoid_strike.py:307 calls): a = v (2 w^2 - 1), b = 2 w (q x v), c = 2 q (q . v)."""
    q_w, q_vec = q[:, -1], q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(q.shape[0], 1, 3), v.view(q.shape[0], 3, 1)).squeeze(-1) * 2.0
    return a + b + c


def strike_reward(tar_pos, tar_rot, root_state, prev_root_pos, strike_body_vel, dt, near_dist):
    up = torch.zeros_like(tar_pos)
    up[..., -1] = 1
    tar_rot_err = torch.sum(up * _quat_rotate_isaac(tar_rot, up), dim=-1)
    tar_rot_r = torch.clamp_min(1.0 - tar_rot_err, 0.0)
    root_pos = root_state[..., 0:3]
    tar_dir = torch.nn.functional.normalize(tar_pos[..., 0:2] - root_pos[..., 0:2], dim=-1)
    root_vel = (root_pos - prev_root_pos) / dt
    tar_dir_speed = torch.sum(tar_dir * root_vel[..., :2], dim=-1)
    tar_vel_err = torch.clamp_min(1.0 - tar_dir_speed, 0.0)
    vel_reward = torch.exp(-4.0 * (tar_vel_err * tar_vel_err))
    vel_reward[tar_dir_speed <= 0] = 0
    reward = 0.6 * tar_rot_r + 0.4 * vel_reward
    return torch.where(tar_rot_err < 0.2, torch.ones_like(reward), reward)


def strike_reset(reset_buf, progress_buf, contact_buf, contact_body_ids, rigid_body_pos, tar_contact_forces, strike_body_ids,
                 max_episode_length, enable_early_termination, termination_heights):
    terminated = torch.zeros_like(reset_buf)
    if enable_early_termination:
        has_fallen, masked = _fall(contact_buf, contact_body_ids, rigid_body_pos, termination_heights)
        tar_has_contact = torch.any(torch.abs(tar_contact_forces[..., 0:2]) > 50.0, dim=-1)
        masked[:, strike_body_ids, :] = 0
        nonstrike = torch.any(torch.any(torch.abs(masked) > 50.0, dim=-1), dim=-1)
        has_failed = torch.logical_or(has_fallen, torch.logical_and(tar_has_contact, nonstrike))
        has_failed = has_failed * (progress_buf > 1)
        terminated = torch.where(has_failed, torch.ones_like(reset_buf), terminated)
    reset = torch.where(progress_buf >= max_episode_length - 1, torch.ones_like(reset_buf), terminated)
    return reset, terminated


# ---------------------------------------------------------------------------------------------------------------------
# HumanoidTraj / HumanoidPedestrianTerrain (phc/env/tasks/humanoid_traj.py, humanoid_pedestrian_terrain.py, phc/utils/traj_generator.py):
# trajectory following over a height field -- the README's terrain-traversal command (learning=pulse_z_terrain, network amp_sept).
#   traj_generate / traj_from_draws   TrajGenerator.reset   phc/utils/traj_generator.py:60-123
#   traj_calc_pos                TrajGenerator.calc_pos     phc/utils/traj_generator.py:156-171
#   fetch_traj_samples           HumanoidTraj._fetch_traj_samples   humanoid_traj.py:196-211
#   traj_location_observations   compute_location_observations      humanoid_pedestrian_terrain.py:1587-1616
#   location_reward(_fuzzy)      compute_location_reward(_fuzzy)    :1619-1646
#   terrain_reset / traj_reset   compute_humanoid_reset             :1476-1531 / humanoid_traj.py:256-300
#   sample_heights               Terrain.world_points_to_map + sample_height_points (no groups)   :1191-1270
#   terrain_heights / terrain_center_heights   get_heights / get_center_heights   :690-772
#   terrain_task_obs             _compute_task_obs          :384-440
# ---------------------------------------------------------------------------------------------------------------------
import math


def traj_generate(init_pos, num_verts, dt, dtheta_max, speed_min, speed_max, accel_max, sharp_turn_prob, generator=None):
    """TrajGenerator.reset for every env in ``init_pos`` (n, 3); draws exactly as the reference does.  Returns verts (n, num_verts, 3)."""
    n = init_pos.shape[0]
    dtheta = 2 * torch.rand([n, num_verts - 1], generator=generator) - 1.0
    dtheta *= dtheta_max * dt
    dtheta_sharp = math.pi * (2 * torch.rand([n, num_verts - 1], generator=generator) - 1.0)
    sharp_probs = sharp_turn_prob * torch.ones_like(dtheta)
    sharp_mask = torch.bernoulli(sharp_probs, generator=generator) == 1.0
    dtheta[sharp_mask] = dtheta_sharp[sharp_mask]
    dtheta[:, 0] = math.pi * (2 * torch.rand([n], generator=generator) - 1.0)
    dspeed = 2 * torch.rand([n, num_verts - 1], generator=generator) - 1.0
    dspeed *= accel_max * dt
    dspeed[:, 0] = (speed_max - speed_min) * torch.rand([n], generator=generator) + speed_min
    return traj_from_draws(init_pos, dtheta, dspeed, dt, speed_min, speed_max)


def traj_from_draws(init_pos, dtheta, dspeed, dt, speed_min, speed_max):
    """The deterministic half of TrajGenerator.reset (:78-113): speed scan with clipping, heading cumsum, vertex cumsum."""
    n, segs = dtheta.shape
    speed = torch.zeros_like(dspeed)
    speed[:, 0] = dspeed[:, 0]
    for i in range(1, segs):
        speed[:, i] = torch.clip(speed[:, i - 1] + dspeed[:, i], speed_min, speed_max)
    theta = torch.cumsum(dtheta, dim=-1)
    seg_len = speed * dt
    dpos = torch.stack([torch.cos(theta), -torch.sin(theta), torch.zeros_like(theta)], dim=-1)
    dpos = dpos * seg_len.unsqueeze(-1)
    dpos[..., 0, 0:2] += init_pos[..., 0:2]
    verts = torch.zeros(n, segs + 1, 3)
    verts[:, 0, 0:2] = init_pos[..., 0:2]
    verts[:, 1:] = torch.cumsum(dpos, dim=-2)
    return verts


def traj_calc_pos(verts, traj_ids, times, dt):
    num_verts = verts.shape[1]
    traj_dur = num_verts * dt                                   # get_traj_duration (:150-153)
    num_segs = num_verts - 1
    phase = torch.clip(times / traj_dur, 0.0, 1.0)
    seg_idx = phase * num_segs
    id0, id1 = torch.floor(seg_idx).long(), torch.ceil(seg_idx).long()
    lerp = (seg_idx - id0).unsqueeze(-1)
    flat = verts.reshape(-1, 3)
    pos0, pos1 = flat[traj_ids * num_verts + id0], flat[traj_ids * num_verts + id1]
    return (1.0 - lerp) * pos0 + lerp * pos1


def fetch_traj_samples(verts, progress, step_dt, traj_dt, num_samples, sample_timestep):
    n = progress.shape[0]
    beg = progress * step_dt
    ts = torch.arange(num_samples, dtype=torch.float) * sample_timestep
    tt = beg.unsqueeze(-1) + ts
    ids = torch.arange(n).unsqueeze(-1).expand(-1, num_samples)
    return traj_calc_pos(verts, ids.flatten(), tt.flatten(), traj_dt).reshape(n, num_samples, 3)


def _base_rot(q, upright):
    from .env_oracle import remove_base_rot
    return q if upright else remove_base_rot(q)


def traj_location_observations(root_states, traj_samples, upright=True):
    root_pos, root_rot = root_states[:, 0:3], _base_rot(root_states[:, 3:7], upright)
    h = R.heading_q_inv(root_rot)
    n, t = traj_samples.shape[0], traj_samples.shape[1]
    he = h.unsqueeze(-2).expand(n, t, 4).reshape(n * t, 4)
    delta = (traj_samples - root_pos.unsqueeze(-2)).reshape(n * t, 3)
    return R.qrot(he, delta)[..., 0:2].reshape(n, t * 2)


def location_reward(root_pos, tar_pos, fuzzy=False):
    d = tar_pos[..., 0:2] - root_pos[..., 0:2]
    err = torch.sum(d * d, dim=-1)
    if fuzzy:
        err = err.clone()
        err[err < 0.0025] = 0
    return torch.exp(-2.0 * err)


def traj_reset(reset_buf, progress_buf, contact_buf, contact_body_ids, rigid_body_pos, tar_pos, max_episode_length, fail_dist,
               enable_early_termination, termination_heights):
    terminated = torch.zeros_like(reset_buf)
    if enable_early_termination:
        has_fallen, _ = _fall(contact_buf, contact_body_ids, rigid_body_pos, termination_heights)
        has_fallen = has_fallen * (progress_buf > 1)
        d = tar_pos[..., 0:2] - rigid_body_pos[..., 0, 0:2]
        tar_fail = torch.sum(d * d, dim=-1) > fail_dist * fail_dist
        terminated = torch.where(torch.logical_or(has_fallen, tar_fail), torch.ones_like(reset_buf), terminated)
    reset = torch.where(progress_buf >= max_episode_length - 1, torch.ones_like(reset_buf), terminated)
    return reset, terminated


def terrain_reset(reset_buf, progress_buf, contact_buf, contact_body_ids, rigid_body_pos, tar_pos, max_episode_length, fail_dist,
                  enable_early_termination, disable_collision=False):
    """The terrain variant: a fall is a summed non-foot contact force above 50 N (no height test)."""
    terminated = torch.zeros_like(reset_buf)
    if enable_early_termination:
        masked = contact_buf.clone()
        masked[:, contact_body_ids, :] = 0
        has_fallen = torch.sqrt(torch.square(torch.abs(masked.sum(dim=-2))).sum(dim=-1)) > 50
        has_fallen = has_fallen * (progress_buf > 1)
        d = tar_pos[..., 0:2] - rigid_body_pos[..., 0, 0:2]
        tar_fail = torch.sum(d * d, dim=-1) > fail_dist * fail_dist
        failed = torch.logical_or(has_fallen, tar_fail)
        if disable_collision:
            failed = torch.zeros_like(failed)
        terminated = torch.where(failed, torch.ones_like(reset_buf), terminated)
    reset = torch.where(progress_buf >= max_episode_length - 1, torch.ones_like(reset_buf), terminated)
    return reset, terminated


def quat_apply(a, b):
    """isaacgym.torch_utils.quat_apply (3P, oracle/shim): b + w t + xyz x t with t = 2 xyz x b."""
    xyz = a[..., :3]
    t = torch.cross(xyz, b, dim=-1) * 2
    return b + a[..., 3:] * t + torch.cross(xyz, t, dim=-1)


def sample_heights(heightsamples, points, horizontal_scale, vertical_scale):
    """Terrain.world_points_to_map + sample_height_points without groups: integer cell (truncation), clip, min of the cell and its
    diagonal neighbour, scaled.  heightsamples: (rows, cols) int16; points (B, P, 3)."""
    p = (points / horizontal_scale).long()
    px = torch.clip(p[:, :, 0].reshape(-1), 0, heightsamples.shape[0] - 2)
    py = torch.clip(p[:, :, 1].reshape(-1), 0, heightsamples.shape[1] - 2)
    h = torch.min(heightsamples[px, py], heightsamples[px + 1, py + 1])
    return (h * vertical_scale).view(points.shape[0], -1)


def terrain_sample_points(sensor_states, height_points, upright=True):
    """World positions of the sensor grid (get_heights, first half): rotated by the HEADING of the sensor body and moved to it."""
    q = _base_rot(sensor_states[:, 3:7], upright)
    h = R.heading_q(q)
    n, p = sensor_states.shape[0], height_points.shape[0]
    pts = quat_apply(h.unsqueeze(1).expand(n, p, 4).reshape(-1, 4), height_points.unsqueeze(0).expand(n, p, 3).reshape(-1, 3)).view(n, p, 3)
    return pts + sensor_states[:, :3].unsqueeze(1)


def terrain_heights(heightsamples, sensor_states, height_points, horizontal_scale, vertical_scale, upright=True):
    """get_heights: the sensor grid rotated by the HEADING of the sensor body (head or root) and moved to it."""
    return sample_heights(heightsamples, terrain_sample_points(sensor_states, height_points, upright), horizontal_scale, vertical_scale)


def terrain_center_heights(heightsamples, root_states, center_points, horizontal_scale, vertical_scale, upright=True):
    """get_center_heights: the 3 x 3 grid under the root, rotated by the root's YAW-ONLY quaternion (quat_apply_yaw)."""
    q = _base_rot(root_states[:, 3:7], upright).clone()
    q[:, :2] = 0.0
    q = q / q.norm(p=2, dim=-1).clamp(min=1e-9).unsqueeze(-1)
    n, p = root_states.shape[0], center_points.shape[0]
    pts = quat_apply(q.unsqueeze(1).expand(n, p, 4).reshape(-1, 4), center_points.unsqueeze(0).expand(n, p, 3).reshape(-1, 3)).view(n, p, 3)
    pts = pts + root_states[:, :3].unsqueeze(1)
    return sample_heights(heightsamples, pts, horizontal_scale, vertical_scale)


def terrain_task_obs(root_states, sensor_states, traj_samples, heightsamples, height_points, center_points, horizontal_scale=0.1,
                     vertical_scale=0.005, upright=True, use_center_height=True, height_meas_scale=5.0):
    obs = traj_location_observations(root_states, traj_samples, upright)
    measured = terrain_heights(heightsamples, sensor_states, height_points, horizontal_scale, vertical_scale, upright)
    if use_center_height:
        center = terrain_center_heights(heightsamples, root_states, center_points, horizontal_scale, vertical_scale, upright).mean(dim=-1, keepdim=True)
        heights = torch.clip(center - measured, -3, 3.) * height_meas_scale
    else:
        heights = torch.clip(root_states[:, 2:3] - measured, -3, 3.) * height_meas_scale
    return torch.cat([obs, heights], dim=1)
