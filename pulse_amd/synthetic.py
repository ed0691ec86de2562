"""Deterministic synthetic inputs for the PULSE hot path.

There is no AMASS data and no Isaac Gym here, so every rollout input is
synthetic: a "physics state" per step (the (N, 24, 13) rigid-body records Isaac
Gym would expose, phc/env/tasks/humanoid.py:215-222), the reference-motion frame
at time t (reward / reset) and t+1 (task observation), dof forces / velocities,
progress counters and episode-end flags.

Generation is on CPU with an explicit ``torch.Generator`` (seed 1234 + rank) so
that the CPU oracle and the HIP path consume bit-identical buffers; callers move
the result to the device.  This module is input plumbing, not part of the
measured path.
"""
import math

import torch

NUM_BODIES = 24          # SMPL_MUJOCO_NAMES, phc/env/tasks/humanoid.py:375-379
NUM_DOF = 69             # 23 joints x 3 exp-map dofs, humanoid.py:643-646
RB_WIDTH = 13            # pos 3, rot xyzw 4, lin vel 3, ang vel 3

SMPL_BODY_NAMES = ['Pelvis', 'L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe', 'R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe',
                   'Torso', 'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist',
                   'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']
# env_im.yaml:38 reset_bodies (every body except the ankles / toes)
RESET_BODY_NAMES = ['Pelvis', 'L_Hip', 'L_Knee', 'R_Hip', 'R_Knee', 'Torso', 'Spine', 'Chest', 'Neck', 'Head',
                    'L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder',
                    'R_Elbow', 'R_Wrist', 'R_Hand']
RESET_BODY_IDS = [SMPL_BODY_NAMES.index(n) for n in RESET_BODY_NAMES]
VR_TRACK_BODY_IDS = [SMPL_BODY_NAMES.index(n) for n in ('Head', 'L_Hand', 'R_Hand')]  # env_pulse_im.yaml:71-72


def make_generator(seed=1234, rank=0):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + rank)
    return g


def _randn(g, *shape):
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def _rand(g, *shape):
    return torch.rand(*shape, generator=g, dtype=torch.float32)


def _quat_mul_xyzw(a, b):
    # plain 16-multiply Hamilton product; only used to CONSTRUCT inputs
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def rigid_body_state(g, n, j=NUM_BODIES):
    """(n, j, 13) records: root xy ~ N(0,1), root z ~ U(.75,1), others root + N(0,.3^2)."""
    rb = torch.empty(n, j, RB_WIDTH, dtype=torch.float32)
    root = torch.cat([_randn(g, n, 2), 0.75 + 0.25 * _rand(g, n, 1)], dim=-1)
    pos = root[:, None, :] + 0.3 * _randn(g, n, j, 3)
    pos[:, 0] = root
    rb[..., 0:3] = pos
    q = _randn(g, n, j, 4)
    rb[..., 3:7] = q / q.norm(dim=-1, keepdim=True)
    rb[..., 7:10] = _randn(g, n, j, 3)
    rb[..., 10:13] = 2.0 * _randn(g, n, j, 3)
    return rb


def reference_frame(g, rb, pos_sigma=0.05, max_angle=0.3, vel_sigma=0.1):
    """A reference-motion frame near ``rb``: dict pos/rot/vel/ang, each (n, j, .)."""
    n, j, _ = rb.shape
    ang = max_angle * _rand(g, n, j)
    ax = _randn(g, n, j, 3)
    ax = ax / ax.norm(dim=-1, keepdim=True)
    dq = torch.cat([ax * torch.sin(0.5 * ang)[..., None], torch.cos(0.5 * ang)[..., None]], dim=-1)
    rot = _quat_mul_xyzw(dq, rb[..., 3:7])
    rot = rot / rot.norm(dim=-1, keepdim=True)
    return {
        "pos": (rb[..., 0:3] + pos_sigma * _randn(g, n, j, 3)).contiguous(),
        "rot": rot.contiguous(),
        "vel": (rb[..., 7:10] + vel_sigma * _randn(g, n, j, 3)).contiguous(),
        "ang": (rb[..., 10:13] + vel_sigma * _randn(g, n, j, 3)).contiguous(),
    }


def env_step_inputs(g, n, edge_cases=True):
    """All inputs of one HumanoidIm.post_physics_step for n envs."""
    rb = rigid_body_state(g, n)
    ref_now = reference_frame(g, rb)
    ref_next = reference_frame(g, rb)
    dof_force = 50.0 * _randn(g, n, NUM_DOF)
    dof_vel = _randn(g, n, NUM_DOF)
    progress = torch.randint(0, 300, (n,), generator=g, dtype=torch.int64)
    pass_time = _rand(g, n) < 0.02
    if edge_cases and n >= 8:
        # identical cur / ref rotation: w rounds to >= 1 -> NaN-mask branch of quat_to_angle_axis
        ref_now["rot"][0] = rb[0, :, 3:7]
        ref_next["rot"][0] = rb[0, :, 3:7]
        # antipodal quaternions (q vs -q)
        ref_now["rot"][1] = -rb[1, :, 3:7]
        ref_next["rot"][1] = -rb[1, :, 3:7]
        # progress <= 1 (no termination) and <= 3 (no power reward)
        progress[2] = 0
        progress[3] = 1
        progress[4] = 3
        progress[5] = 4
        # guaranteed far-away body -> termination when progress > 1
        ref_now["pos"][5, 13] += 1.0
        ref_now["pos"][3, 9] += 1.0      # far, but progress <= 1 -> not terminated
        # identity root rotation (heading 0) and a pure-yaw root
        rb[6, 0, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0])
        rb[7, 0, 3:7] = torch.tensor([0.0, 0.0, math.sin(0.6), math.cos(0.6)])
        pass_time[2] = True
    return {"rb": rb, "ref_now": ref_now, "ref_next": ref_next, "dof_force": dof_force,
            "dof_vel": dof_vel, "progress": progress, "pass_time": pass_time}


def rollout_scalars(g, t, n, done_p=0.02):
    """Stand-alone GAE inputs: rewards / values / next_values (t,n,1), dones (t,n) u8."""
    rewards = _rand(g, t, n, 1)
    values = _randn(g, t, n, 1)
    next_values = _randn(g, t, n, 1)
    dones = (_rand(g, t, n) < done_p)
    term = dones & (_rand(g, t, n) < 0.5)
    next_values = next_values * (1.0 - term.float()[..., None])
    return rewards, values, next_values, dones.to(torch.uint8)
