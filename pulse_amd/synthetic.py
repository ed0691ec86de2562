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


# ---------------------------------------------------------------------------------------------------------------------
# Synthetic motion library: the flat per-frame tables MotionLibBase builds from AMASS clips
# (phc/utils/motion_lib_base.py:287-316): gts/grs/lrs/gvs/gavs (F, 24, 3|4), dvs (F, 23, 3) plus per-motion
# length / fps / dt / frame count.  Clips are smooth random articulated motions of an SMPL-shaped kinematic tree
# (forward kinematics over SMPL_PARENTS), velocities by finite differences like SkeletonMotion does.
# ---------------------------------------------------------------------------------------------------------------------
SMPL_PARENTS = [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]


def _exp_map_to_quat_xyzw(e):
    ang = e.norm(dim=-1, keepdim=True)
    half = 0.5 * ang
    k = torch.where(ang > 1e-8, torch.sin(half) / ang.clamp_min(1e-8), torch.full_like(ang, 0.5))
    return torch.cat([e * k, torch.cos(half)], dim=-1)


def _quat_rotate_xyzw(q, v):
    qv, w = q[..., :3], q[..., 3:4]
    t = 2.0 * torch.cross(qv, v, dim=-1)
    return v + w * t + torch.cross(qv, t, dim=-1)


def synthetic_motion_library(g, num_motions, min_frames=45, max_frames=180, fps=30.0):
    """dict of CPU tensors in the reference's table layout; frame f of motion m sits at length_starts[m] + f."""
    m, j = num_motions, NUM_BODIES
    num_frames = torch.randint(min_frames, max_frames + 1, (m,), generator=g, dtype=torch.int64)
    starts = torch.cumsum(num_frames, 0) - num_frames
    total = int(num_frames.sum())
    mid = torch.repeat_interleave(torch.arange(m), num_frames)                     # motion of every frame
    fidx = torch.arange(total) - starts[mid]
    dt = 1.0 / fps
    t = (fidx.float() * dt)[:, None, None]                                         # (F,1,1)
    # joint exp-map trajectories: two sinusoids per dof
    amp = 0.35 * _rand(g, m, j, 3, 2)
    amp[:, 0] *= 0.3                                                               # pelvis tilt stays small
    frq = 0.3 + 1.7 * _rand(g, m, j, 3, 2)
    pha = 6.2831853 * _rand(g, m, j, 3, 2)
    e = (amp[mid] * torch.sin(6.2831853 * frq[mid] * t[..., None] + pha[mid])).sum(-1)      # (F,24,3)
    yaw_rate = 0.8 * _randn(g, m)
    e[:, 0, 2] += 6.2831853 * _rand(g, m)[mid] + yaw_rate[mid] * t[:, 0, 0]
    lrs = _exp_map_to_quat_xyzw(e)
    lrs = lrs / lrs.norm(dim=-1, keepdim=True)
    offsets = 0.08 + 0.3 * _rand(g, j, 3) * torch.tensor([0.4, 0.4, 1.0])
    offsets[0] = 0.0
    # root path: constant drift + sway
    drift = 0.6 * _randn(g, m, 3) * torch.tensor([1.0, 1.0, 0.0])
    sway = 0.05 * _randn(g, m, 3)
    root = drift[mid] * t[:, 0] + sway[mid] * torch.sin(3.0 * t[:, 0]) + torch.tensor([0.0, 0.0, 0.9])
    grs = torch.empty(total, j, 4)
    gts = torch.empty(total, j, 3)
    for b in range(j):
        p = SMPL_PARENTS[b]
        if p < 0:
            grs[:, b], gts[:, b] = lrs[:, b], root
        else:
            grs[:, b] = _quat_mul_xyzw(grs[:, p], lrs[:, b])
            gts[:, b] = gts[:, p] + _quat_rotate_xyzw(grs[:, p], offsets[b].expand(total, 3))
    grs = grs / grs.norm(dim=-1, keepdim=True)

    def fdiff(x):                                                                  # forward difference inside each clip, last frame repeats
        nxt = torch.roll(x, -1, 0)
        d = (nxt - x) / dt
        last = fidx == (num_frames[mid] - 1)
        d[last] = d[torch.where(last)[0] - 1]
        return d

    def ang_vel(q):                                                                # 2 (q_{f+1} conj(q_f)).xyz / dt
        nxt = torch.roll(q, -1, 0)
        conj = q * torch.tensor([-1.0, -1.0, -1.0, 1.0])
        dq = _quat_mul_xyzw(nxt, conj)
        dq = torch.where(dq[..., 3:4] < 0, -dq, dq)
        w = 2.0 * dq[..., :3] / dt
        last = fidx == (num_frames[mid] - 1)
        w[last] = w[torch.where(last)[0] - 1]
        return w

    gvs, gavs = fdiff(gts), ang_vel(grs)
    dvs = ang_vel(lrs)[:, 1:]
    return {
        "gts": gts.contiguous(), "grs": grs.contiguous(), "lrs": lrs.contiguous(), "gvs": gvs.contiguous(),
        "gavs": gavs.contiguous(), "dvs": dvs.contiguous(),
        "motion_num_frames": num_frames, "motion_fps": torch.full((m,), fps), "motion_dt": torch.full((m,), dt),
        "motion_lengths": ((1.0 / fps) * (num_frames - 1).double()).float(), "length_starts": starts,   # curr_len, motion_lib_base.py:263
    }


# --------------------------------------------------------------------------- #
# Action-dependent physics stand-in (return-parity experiments): constants shared by the HIP kernel's host class
# (pulse_amd/env/sim.py:PdSim) and its CPU twin (oracle/pd_sim_oracle.py).  See include/pulse_hip.h: pulse_pd_sim_args.
# --------------------------------------------------------------------------- #
PD_SIM = {"kp": 400.0, "kd": 40.0, "substeps": 2, "action_scale": 0.25, "lever": 0.3, "noise_acc": 2.0}


def pd_sim_tables():
    """(sag (69,), lever_dir (24, 3)): a constant per-joint offset of the PD set point the policy has to learn to cancel, and the unit
    'bone' directions that turn a joint-angle error into a body displacement.  Deterministic (no RNG state involved)."""
    j = torch.arange(NUM_DOF, dtype=torch.float32)
    sag = 0.25 * torch.sin(0.7 * (j + 1.0))
    b = torch.arange(NUM_BODIES, dtype=torch.float32)
    u = torch.stack([torch.cos(1.3 * b), torch.sin(1.3 * b) * torch.cos(0.9 * b + 0.4), torch.sin(1.3 * b) * torch.sin(0.9 * b + 0.4)], dim=-1)
    return sag, u / u.norm(dim=-1, keepdim=True)


def synthetic_height_field(rows=260, cols=300, seed=5):
    """A height field with slopes, steps and noise in Isaac Gym's storage format (int16 samples, vertical scale 0.005 m, cells of 0.1 m):
    the stand-in for Terrain.heightsamples (phc/env/tasks/humanoid_pedestrian_terrain.py:1114-1160 builds it with isaacgym.terrain_utils,
    a closed third-party package)."""
    g = torch.Generator().manual_seed(seed)
    x, y = torch.meshgrid(torch.arange(rows).float(), torch.arange(cols).float(), indexing="ij")
    h = 0.6 * torch.sin(x / 23.0) * torch.cos(y / 31.0) + 0.15 * torch.floor(x / 40.0) + 0.05 * torch.randn(rows, cols, generator=g)
    return torch.round(h / 0.005).to(torch.int16)


def square_height_points(extent=2.0, res=32):
    """init_square_height_points (humanoid_pedestrian_terrain.py:608-625): res x res grid over [-extent, extent]^2, x-major -> (res^2, 2)."""
    import numpy as np
    x = torch.tensor(np.linspace(-extent, extent, res))
    gx, gy = torch.meshgrid(x, x.clone(), indexing="ij")
    return torch.stack([gx.flatten(), gy.flatten()], dim=1).float()


def center_height_points():
    """init_center_height_points (:591-606): x in linspace(-0.1, 0.1, 3), y in linspace(-0.2, 0.2, 3) -> (9, 2)."""
    import numpy as np
    x, y = torch.tensor(np.linspace(-0.1, 0.1, 3)), torch.tensor(np.linspace(-0.2, 0.2, 3))
    gx, gy = torch.meshgrid(x, y, indexing="ij")
    return torch.stack([gx.flatten(), gy.flatten()], dim=1).float()
