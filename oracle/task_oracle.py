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
