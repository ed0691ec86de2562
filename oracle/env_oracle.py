"""PyTorch-CPU restatement of the env-side hot path (obs / reward / reset / GAE).

ORACLE / TEST INFRASTRUCTURE -- never imported by ``pulse_amd``.

Shapes: N envs, J=24 SMPL bodies, Jt tracked bodies, T future samples.
``rb`` is Isaac's rigid-body record (N, J, 13): pos 0:3, rot xyzw 3:7,
lin vel 7:10, ang vel 10:13 (phc/env/tasks/humanoid.py:215-222).

Pinned by tests/golden/env_*.npz (outputs of the reference's own jit functions,
AST-loaded by oracle/refload.py and run in the build container).
"""
import torch

from . import rotations as R


def split_rb(rb):
    return rb[..., 0:3], rb[..., 3:7], rb[..., 7:10], rb[..., 10:13]


def self_obs_smpl_max(body_pos, body_rot, body_vel, body_ang_vel,
                      local_root_obs=True, root_height_obs=True):
    """compute_humanoid_observations_smpl_max, phc/env/tasks/humanoid.py:1675-1731
    (upright start, no shape / limb-weight params) -> (N, 1 + 15 J - 3)."""
    n, j, _ = body_pos.shape
    root_pos = body_pos[:, 0, :]
    root_rot = body_rot[:, 0, :]
    h_inv = R.heading_q_inv(root_rot)
    h_inv_flat = h_inv.unsqueeze(-2).repeat((1, j, 1)).reshape(n * j, 4)

    rel = (body_pos - root_pos.unsqueeze(-2)).reshape(n * j, 3)
    loc_pos = R.qrot(h_inv_flat, rel).reshape(n, j * 3)[..., 3:]

    loc_rot = R.qmul(h_inv_flat, body_rot.reshape(n * j, 4))
    rot6 = R.q_to_tan_norm(loc_rot).reshape(n, j * 6)
    if not local_root_obs:
        rot6[..., 0:6] = R.q_to_tan_norm(root_rot)

    loc_vel = R.qrot(h_inv_flat, body_vel.reshape(n * j, 3)).reshape(n, j * 3)
    loc_ang = R.qrot(h_inv_flat, body_ang_vel.reshape(n * j, 3)).reshape(n, j * 3)

    parts = []
    if root_height_obs:
        parts.append(root_pos[:, 2:3])
    parts += [loc_pos, rot6, loc_vel, loc_ang]
    return torch.cat(parts, dim=-1)


def im_obs_v6(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
              ref_pos, ref_rot, ref_vel, ref_ang_vel, time_steps=1):
    """compute_imitation_observations_v6, phc/env/tasks/humanoid_im.py:1328-1378
    (upright start).  cur (N,Jt,.), ref (N*T,Jt,.) -> (N, 24*Jt*T)."""
    b, j, _ = body_pos.shape
    t = time_steps
    h_inv = R.heading_q_inv(root_rot)
    h = R.heading_q(root_rot)
    h_inv_e = h_inv.unsqueeze(-2).repeat((1, j, 1)).repeat_interleave(t, 0).view(-1, 4)
    h_e = h.unsqueeze(-2).repeat((1, j, 1)).repeat_interleave(t, 0).view(-1, 4)

    d_pos = ref_pos.view(b, t, j, 3) - body_pos.view(b, 1, j, 3)
    d_pos_loc = R.qrot(h_inv_e, d_pos.view(-1, 3))

    cur_rot_e = body_rot[:, None].repeat_interleave(t, 1)
    d_rot = R.qmul(ref_rot.view(b, t, j, 4), R.qconj(cur_rot_e))
    d_rot_loc = R.qmul(R.qmul(h_inv_e, d_rot.view(-1, 4)), h_e)

    d_vel = ref_vel.view(b, t, j, 3) - body_vel.view(b, 1, j, 3)
    d_vel_loc = R.qrot(h_inv_e, d_vel.view(-1, 3))
    d_ang = ref_ang_vel.view(b, t, j, 3) - body_ang_vel.view(b, 1, j, 3)
    d_ang_loc = R.qrot(h_inv_e, d_ang.view(-1, 3))

    ref_rel = ref_pos.view(b, t, j, 3) - root_pos.view(b, 1, 1, 3)
    ref_rel_loc = R.qrot(h_inv_e, ref_rel.view(-1, 3))
    ref_rot_loc6 = R.q_to_tan_norm(R.qmul(h_inv_e, ref_rot.view(-1, 4)))

    blocks = [d_pos_loc.view(b, t, -1), R.q_to_tan_norm(d_rot_loc).view(b, t, -1),
              d_vel_loc.view(b, t, -1), d_ang_loc.view(b, t, -1),
              ref_rel_loc.view(b, t, -1), ref_rot_loc6.view(b, t, -1)]
    return torch.cat(blocks, dim=-1).view(b, -1)


def im_obs_v7(root_pos, root_rot, body_pos, body_vel, ref_pos, ref_vel, time_steps=1):
    """compute_imitation_observations_v7, phc/env/tasks/humanoid_im.py:1381-1413
    -> (N, 9*Jt*T)."""
    b, j, _ = body_pos.shape
    t = time_steps
    h_inv_e = R.heading_q_inv(root_rot).unsqueeze(-2).repeat((1, j, 1)).repeat_interleave(t, 0).view(-1, 4)
    d_pos = R.qrot(h_inv_e, (ref_pos.view(b, t, j, 3) - body_pos.view(b, 1, j, 3)).view(-1, 3))
    d_vel = R.qrot(h_inv_e, (ref_vel.view(b, t, j, 3) - body_vel.view(b, 1, j, 3)).view(-1, 3))
    rel = R.qrot(h_inv_e, (ref_pos.view(b, t, j, 3) - root_pos.view(b, 1, 1, 3)).view(-1, 3))
    return torch.cat([d_pos.view(b, t, -1), d_vel.view(b, t, -1), rel.view(b, t, -1)], dim=-1).view(b, -1)


def remove_base_rot(quat):
    """remove_base_rot, phc/env/tasks/humanoid.py:1616-1620: q * conj([0.5, 0.5, 0.5, 0.5]) (SMPL's non-upright rest frame)."""
    base = R.qconj(torch.tensor([[0.5, 0.5, 0.5, 0.5]]).to(quat))
    return R.qmul(quat, base.repeat(quat.shape[0], 1))


def _im_blocks(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel, t, upright):
    """The eight per-(sample, body) blocks every compute_imitation_observations* variant is assembled from
    (phc/env/tasks/humanoid_im.py:1222-1540), each flattened (B*T*J, .)."""
    b, j, _ = body_pos.shape
    if not upright:
        root_rot = remove_base_rot(root_rot)
    h_inv = R.heading_q_inv(root_rot)
    h = R.heading_q(root_rot)
    h_inv_e = h_inv.unsqueeze(-2).repeat((1, j, 1)).repeat_interleave(t, 0).view(-1, 4)
    h_e = h.unsqueeze(-2).repeat((1, j, 1)).repeat_interleave(t, 0).view(-1, 4)
    d_pos = R.qrot(h_inv_e, (ref_pos.view(b, t, j, 3) - body_pos.view(b, 1, j, 3)).view(-1, 3))
    d_rot = R.qmul(ref_rot.view(b, t, j, 4), R.qconj(body_rot[:, None].repeat_interleave(t, 1)))
    d_rot6 = R.q_to_tan_norm(R.qmul(R.qmul(h_inv_e, d_rot.view(-1, 4)), h_e))
    d_vel = R.qrot(h_inv_e, (ref_vel.view(b, t, j, 3) - body_vel.view(b, 1, j, 3)).view(-1, 3))
    d_ang = R.qrot(h_inv_e, (ref_ang_vel.view(b, t, j, 3) - body_ang_vel.view(b, 1, j, 3)).view(-1, 3))
    r_pos = R.qrot(h_inv_e, (ref_pos.view(b, t, j, 3) - root_pos.view(b, 1, 1, 3)).view(-1, 3))
    r_rot6 = R.q_to_tan_norm(R.qmul(h_inv_e, ref_rot.view(-1, 4)))
    r_vel = R.qrot(h_inv_e, ref_vel.view(-1, 3))
    r_ang = R.qrot(h_inv_e, ref_ang_vel.view(-1, 3))
    return {"d_pos": d_pos, "d_rot6": d_rot6, "d_vel": d_vel, "d_ang": d_ang, "r_pos": r_pos, "r_rot6": r_rot6, "r_vel": r_vel,
            "r_ang": r_ang, "h_inv": h_inv}


def im_obs_variant(version, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel,
                   time_steps=1, upright=True, dof_pos=None, ref_dof_pos=None):
    """compute_imitation_observations (v1, :1222-1256), _v2 (:1259-1297, + dof differences), _v3 (:1300-1325, no velocities),
    _v6 / _v7 (with ``upright``), _v8 (:1415-1481, one sample: T > 1 slices a column there) and _v9 (:1484-1540, root
    velocity differences only).  cur (B, Jt, .), ref (B*T, Jt, .)."""
    b, j, _ = body_pos.shape
    t = time_steps
    k = _im_blocks(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel, t, upright)
    v = lambda x: x.view(b, -1)
    vt = lambda x: x.view(b, t, -1)
    if version == 1:
        return torch.cat([v(k["d_pos"]), v(k["d_rot6"]), v(k["d_vel"]), v(k["d_ang"])], dim=-1)
    if version == 2:
        d_dof = ref_dof_pos.view(b, t, -1) - dof_pos.view(b, t, -1)
        return torch.cat([v(k["d_pos"]), v(k["d_rot6"]), v(k["d_vel"]), v(k["d_ang"]), v(d_dof)], dim=-1)
    if version == 3:
        return torch.cat([v(k["d_pos"]), v(k["d_rot6"])], dim=-1)
    if version == 6:
        return torch.cat([vt(k["d_pos"]), vt(k["d_rot6"]), vt(k["d_vel"]), vt(k["d_ang"]), vt(k["r_pos"]), vt(k["r_rot6"])], dim=-1).view(b, -1)
    if version == 7:
        return torch.cat([vt(k["d_pos"]), vt(k["d_vel"]), vt(k["r_pos"])], dim=-1).view(b, -1)
    if version == 8:
        assert t == 1, "v8 with future samples indexes a column of the flattened velocities in the reference (:1466-1474)"
        return torch.cat([v(k["d_pos"]), v(k["d_rot6"]), v(k["d_vel"]), v(k["d_ang"]), v(k["r_pos"]), v(k["r_rot6"]), v(k["r_vel"]),
                          v(k["r_ang"])], dim=-1)
    if version == 9:
        h_root = k["h_inv"].repeat_interleave(t, 0)
        d_rv = R.qrot(h_root, (ref_vel.view(b, t, j, 3)[:, :, 0] - body_vel[:, None, 0]).reshape(-1, 3))
        d_ra = R.qrot(h_root, (ref_ang_vel.view(b, t, j, 3)[:, :, 0] - body_ang_vel[:, None, 0]).reshape(-1, 3))
        return torch.cat([vt(k["d_pos"]), vt(k["d_rot6"]), vt(d_rv), vt(d_ra), vt(k["r_pos"]), vt(k["r_rot6"])], dim=-1).view(b, -1)
    raise ValueError(version)


def self_obs_smpl_max_general(body_pos, body_rot, body_vel, body_ang_vel, local_root_obs=True, root_height_obs=True, upright=True,
                              force_sensor=None, smpl_params=None, limb_weight_params=None):
    """compute_humanoid_observations_smpl_max with ``upright`` False (humanoid.py:1675-1731: heading AND the non-local root 6-D block
    from remove_base_rot(root)) and _max_v3 (:1789-1849: the same with the force-sensor readings appended)."""
    n, j, _ = body_pos.shape
    root_pos = body_pos[:, 0, :]
    root_rot = body_rot[:, 0, :]
    hr = root_rot if upright else remove_base_rot(root_rot)
    h_inv_flat = R.heading_q_inv(hr).unsqueeze(-2).repeat((1, j, 1)).reshape(n * j, 4)
    loc_pos = R.qrot(h_inv_flat, (body_pos - root_pos.unsqueeze(-2)).reshape(n * j, 3)).reshape(n, j * 3)[..., 3:]
    rot6 = R.q_to_tan_norm(R.qmul(h_inv_flat, body_rot.reshape(n * j, 4))).reshape(n, j * 6)
    if not local_root_obs:
        rot6[..., 0:6] = R.q_to_tan_norm(hr)
    loc_vel = R.qrot(h_inv_flat, body_vel.reshape(n * j, 3)).reshape(n, j * 3)
    loc_ang = R.qrot(h_inv_flat, body_ang_vel.reshape(n * j, 3)).reshape(n, j * 3)
    parts = [root_pos[:, 2:3]] if root_height_obs else []
    parts += [loc_pos, rot6, loc_vel, loc_ang]
    if force_sensor is not None:
        parts.append(force_sensor)
    if smpl_params is not None:               # has_smpl_params, then has_limb_weight_params (humanoid.py:1724-1728, 1843-1847)
        parts.append(smpl_params)
    if limb_weight_params is not None:
        parts.append(limb_weight_params)
    return torch.cat(parts, dim=-1)


def self_obs_smpl_max_v2(body_pos, body_rot, body_vel, body_ang_vel, local_root_obs=True, root_height_obs=True, upright=True):
    """compute_humanoid_observations_smpl_max_v2, humanoid.py:1734-1786: a history of T simulated states (B, T, J, .), every step
    expressed in the heading frame of the LATEST root, per step [height | pos | rot6 | vel | ang]."""
    b, t, j, _ = body_pos.shape
    root_pos = body_pos[:, -1, 0, :]
    root_rot = body_rot[:, -1, 0, :]
    if not upright:
        root_rot = remove_base_rot(root_rot)
    h_inv_e = R.heading_q_inv(root_rot).unsqueeze(-2).repeat((1, j, 1)).repeat_interleave(t, 0).view(-1, 4)
    loc_pos = R.qrot(h_inv_e, (body_pos - root_pos.unsqueeze(-2).unsqueeze(-2)).view(-1, 3)).reshape(b, t, j * 3)[..., 3:]
    rot6 = R.q_to_tan_norm(R.qmul(h_inv_e, body_rot.view(-1, 4))).view(b, t, j * 6)
    if not local_root_obs:
        raise NotImplementedError("the reference raises here (a (B*T, 6) block assigned into a (B, T, 6) slot, humanoid.py:1766-1768)")
    loc_vel = R.qrot(h_inv_e, body_vel.view(-1, 3)).view(b, t, j * 3)
    loc_ang = R.qrot(h_inv_e, body_ang_vel.view(-1, 3)).view(b, t, j * 3)
    body_obs = torch.cat([loc_pos, rot6, loc_vel, loc_ang], dim=-1)
    if root_height_obs:
        body_obs = torch.cat([body_pos[:, :, 0, 2:3], body_obs], dim=-1)
    return body_obs.view(b, -1)


DEFAULT_REWARD_SPECS = {  # phc/env/tasks/humanoid_im.py:55
    "k_pos": 100.0, "k_rot": 10.0, "k_vel": 0.1, "k_ang_vel": 0.1,
    "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1,
}
DEFAULT_POWER_COEF = 0.0005  # phc/env/tasks/humanoid_im.py:92


def im_reward(body_pos, body_rot, body_vel, body_ang_vel,
              ref_pos, ref_rot, ref_vel, ref_ang_vel, specs=None):
    """compute_imitation_reward, phc/env/tasks/humanoid_im.py:1543-1574."""
    s = DEFAULT_REWARD_SPECS if specs is None else specs
    e_pos = ((ref_pos - body_pos) ** 2).mean(dim=-1).mean(dim=-1)
    r_pos = torch.exp(-s["k_pos"] * e_pos)
    ang = R.q_to_angle_axis(R.qmul(ref_rot, R.qconj(body_rot)))[0]
    r_rot = torch.exp(-s["k_rot"] * (ang ** 2).mean(dim=-1))
    e_vel = ((ref_vel - body_vel) ** 2).mean(dim=-1).mean(dim=-1)
    r_vel = torch.exp(-s["k_vel"] * e_vel)
    e_ang = ((ref_ang_vel - body_ang_vel) ** 2).mean(dim=-1).mean(dim=-1)
    r_ang = torch.exp(-s["k_ang_vel"] * e_ang)
    rew = s["w_pos"] * r_pos + s["w_rot"] * r_rot + s["w_vel"] * r_vel + s["w_ang_vel"] * r_ang
    return rew, torch.stack([r_pos, r_rot, r_vel, r_ang], dim=-1)


def power_term(dof_force, dof_vel, progress, coef=DEFAULT_POWER_COEF):
    """power reward, phc/env/tasks/humanoid_im.py:908-917."""
    p = -coef * torch.abs(torch.multiply(dof_force, dof_vel)).sum(dim=-1)
    p[progress <= 3] = 0
    return p


def im_reward_full(rb, ref_pos, ref_rot, ref_vel, ref_ang_vel, dof_force, dof_vel, progress,
                   specs=None, power_coef=DEFAULT_POWER_COEF, power_reward=True):
    """HumanoidIm._compute_reward with _full_body_reward, humanoid_im.py:853-919."""
    bp, br, bv, ba = split_rb(rb)
    rew, raw = im_reward(bp, br, bv, ba, ref_pos, ref_rot, ref_vel, ref_ang_vel, specs)
    if power_reward:
        p = power_term(dof_force, dof_vel, progress, power_coef)
        rew = rew + p
        raw = torch.cat([raw, p[:, None]], dim=-1)
    return rew, raw


def im_reset(reset_buf, progress, body_pos, ref_pos, pass_time, term_dist,
             enable_early_termination=True, disable_collision=False, use_mean=False):
    """compute_humanoid_im_reset, phc/env/tasks/humanoid_im.py:1600-1628.
    body_pos/ref_pos (N,Jr,3) already restricted to the reset bodies;
    term_dist (1,Jr) or (N,Jr).  Returns (reset, terminated) int64."""
    terminated = torch.zeros_like(reset_buf)
    if enable_early_termination:
        dist = torch.norm(body_pos - ref_pos, dim=-1)
        if use_mean:
            fallen = torch.any(dist.mean(dim=-1, keepdim=True) > term_dist[0], dim=-1)
        else:
            fallen = torch.any(dist > term_dist, dim=-1)
        fallen = fallen * (progress > 1)
        if disable_collision:
            fallen[:] = False
        terminated = torch.where(fallen, torch.ones_like(reset_buf), terminated)
    reset = torch.where(pass_time, torch.ones_like(reset_buf), terminated)
    return reset, terminated


def gae(fdones, values, rewards, next_values, gamma, tau):
    """CommonAgent.discount_values, phc/learning/common_agent.py:493-505.
    All (T,N,1) except fdones (T,N)."""
    last = 0
    advs = torch.zeros_like(rewards)
    for t in reversed(range(rewards.shape[0])):
        nd = (1.0 - fdones[t]).unsqueeze(1)
        delta = rewards[t] + gamma * next_values[t] - values[t]
        last = delta + gamma * tau * nd * last
        advs[t] = last
    return advs


def post_physics(rb, ref_now, ref_next, dof_force, dof_vel, progress, pass_time,
                 reset_body_ids, track_body_ids, term_dist, cycle_counter=None, obs_v=6,
                 specs=None, power_coef=DEFAULT_POWER_COEF, power_reward=True, use_mean=False):
    """One HumanoidIm.post_physics_step after ``progress += 1`` and the sim refresh:
    reward -> reset -> next observation, in the reference's order
    (phc/env/tasks/humanoid.py:1315-1331, humanoid_im.py:677-706, 853-919, 1119-1192).

    ref_now / ref_next: dicts with pos (N,J,3) rot (N,J,4) vel ang (N,J,3) at
    motion time t (reward/reset) and t+1 (task obs).  Returns dict with obs (N,934
    for v6 full body), rew, raw, reset, terminate.
    """
    bp, br, bv, ba = split_rb(rb)
    rew, raw = im_reward_full(rb, ref_now["pos"], ref_now["rot"], ref_now["vel"], ref_now["ang"],
                              dof_force, dof_vel, progress, specs, power_coef, power_reward)
    n = rb.shape[0]
    reset0 = torch.zeros(n, dtype=torch.int64)
    reset, term = im_reset(reset0, progress, bp[:, reset_body_ids].clone(),
                           ref_now["pos"][:, reset_body_ids].clone(), pass_time,
                           term_dist[..., reset_body_ids], use_mean=use_mean)
    if cycle_counter is not None:
        rec = torch.logical_and(~pass_time, cycle_counter > 0)
        reset[rec] = 0
        term[rec] = 0
    so = self_obs_smpl_max(bp, br, bv, ba)
    tb = track_body_ids
    if obs_v == 6:
        to = im_obs_v6(bp[:, 0], br[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb],
                       ref_next["pos"][:, tb], ref_next["rot"][:, tb], ref_next["vel"][:, tb],
                       ref_next["ang"][:, tb], 1)
    elif obs_v == 7:
        to = im_obs_v7(bp[:, 0], br[:, 0], bp[:, tb], bv[:, tb], ref_next["pos"][:, tb],
                       ref_next["vel"][:, tb], 1)
    else:
        raise NotImplementedError(obs_v)
    return {"obs": torch.cat([so, to], dim=-1), "rew": rew, "raw": raw, "reset": reset, "terminate": term}


def dof_to_obs_smpl(pose):
    """dof_to_obs_smpl, phc/env/tasks/humanoid.py:1436-1446: exp-map dofs (B, 3 Jd) -> 6-D rotations (B, 6 Jd)."""
    b = pose.shape[0]
    return R.q_to_tan_norm(R.exp_map_to_q(pose.reshape(-1, 3))).reshape(b, -1)


def amp_obs_smpl(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos, dof_subset=None,
                 local_root_obs=True, root_height_obs=True):
    """build_amp_observations_smpl, phc/env/tasks/humanoid_amp.py:925-969 (upright, no shape / limb obs)."""
    h_inv = R.heading_q_inv(root_rot)
    rot6 = R.q_to_tan_norm(R.qmul(h_inv, root_rot) if local_root_obs else root_rot)
    lvel = R.qrot(h_inv, root_vel)
    lang = R.qrot(h_inv, root_ang_vel)
    rel = key_body_pos - root_pos.unsqueeze(-2)
    nk = rel.shape[1]
    h_e = h_inv.unsqueeze(-2).repeat((1, nk, 1))
    lkey = R.qrot(h_e.view(-1, 4), rel.view(-1, 3)).view(rel.shape[0], nk * 3)
    if dof_subset is not None:
        dof_vel = dof_vel[:, dof_subset]
        dof_pos = dof_pos[:, dof_subset]
    parts = [root_pos[:, 2:3]] if root_height_obs else []
    parts += [rot6, lvel, lang, dof_to_obs_smpl(dof_pos), dof_vel, lkey]
    return torch.cat(parts, dim=-1)


# ---------------------------------------------------------------------------------------------------------------------------
# zero_out_far (humanoid.py:311-329): envs whose root is further than close_distance from the reference see a reference that
# is their own simulated state (plus a direction towards the goal), and are paid for approaching it instead of imitating.
# ---------------------------------------------------------------------------------------------------------------------------
def zero_out_far_refs(obs_v, root_pos, body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel,
                      close_distance=0.25, far_distance=3.0):
    """The zero_out_far blocks of HumanoidIm._compute_task_obs, phc/env/tasks/humanoid_im.py:763-777 (obs_v 4 / 5 / 6 / 8 / 9) and
    :814-826 (obs_v 7: positions and velocities only).  All (N, Jt, .) over the TRACKED bodies, one reference sample.
    Returns the masked copies of the four reference tensors and ``distance`` (= the new _point_goal)."""
    ref_pos, ref_rot, ref_vel, ref_ang_vel = ref_pos.clone(), ref_rot.clone(), ref_vel.clone(), ref_ang_vel.clone()
    distance = torch.norm(root_pos - ref_pos[..., 0, :], dim=-1)
    zeros_subset = distance > close_distance
    ref_pos[zeros_subset, 1:] = body_pos[zeros_subset, 1:]
    if obs_v != 7:
        ref_rot[zeros_subset, 1:] = body_rot[zeros_subset, 1:]
    ref_vel[zeros_subset, :] = body_vel[zeros_subset, :]
    if obs_v != 7:
        ref_ang_vel[zeros_subset, :] = body_ang_vel[zeros_subset, :]
    vector_zero_subset = distance > far_distance                      # beyond far_distance the root target is just a direction
    ref_pos[vector_zero_subset, 0] = ((ref_pos[vector_zero_subset, 0] - body_pos[vector_zero_subset, 0]) / distance[vector_zero_subset, None]
                                      * far_distance) + body_pos[vector_zero_subset, 0]
    return ref_pos, ref_rot, ref_vel, ref_ang_vel, distance


def point_goal_reward(prev_dist, curr_dist):
    """compute_point_goal_reward, phc/env/tasks/humanoid_im.py:1577-1582."""
    reward = torch.clamp(prev_dist - curr_dist, max=1 / 3) * 9
    return reward, reward


def im_reward_zero_out_far(rb, ref_pos, ref_rot, ref_vel, ref_ang_vel, point_goal, dof_force, dof_vel, progress,
                           specs=None, power_coef=DEFAULT_POWER_COEF, power_reward=True):
    """HumanoidIm._compute_reward with zero_out_far, humanoid_im.py:870-887 (+ the power term :908-917): the point-goal reward for
    everyone, plus half the FULL-BODY imitation reward for the envs within 0.25 m (``transition_distance``) of the reference root."""
    bp, br, bv, ba = split_rb(rb)
    n = rb.shape[0]
    distance = torch.norm(bp[..., 0, :] - ref_pos[..., 0, :], dim=-1)          # ref_root_pos = rg_pos[..., 0, :] (motion_lib_base.py:500)
    zeros_subset = distance > 0.25
    raw = torch.zeros((n, 4))
    rew, raw[:, 0] = point_goal_reward(point_goal, distance)
    inside = ~zeros_subset
    im_rew, im_raw = im_reward(bp[inside], br[inside], bv[inside], ba[inside], ref_pos[inside], ref_rot[inside], ref_vel[inside],
                               ref_ang_vel[inside], specs)
    rew[inside] = rew[inside] + im_rew * 0.5
    raw[inside, :4] = raw[inside, :4] + im_raw * 0.5
    if power_reward:
        p = power_term(dof_force, dof_vel, progress, power_coef)
        rew = rew + p
        raw = torch.cat([raw, p[:, None]], dim=-1)
    return rew, raw


def post_physics_zero_out_far(rb, ref_now, ref_next, point_goal, dof_force, dof_vel, progress, pass_time, reset_body_ids, track_body_ids,
                              term_dist, cycle_counter=None, obs_v=6, close_distance=0.25, far_distance=3.0, specs=None,
                              power_coef=DEFAULT_POWER_COEF, power_reward=True, upright=True):
    """post_physics with ``zero_out_far: True`` (phc_kp_pnn_iccv.yaml:36 & co): reward on the old _point_goal, the unchanged reset
    (humanoid_im.py:1158-1176 calls compute_humanoid_im_reset on the un-masked reference), then the task observation against the
    masked reference, which also leaves the new _point_goal.  Returns post_physics' dict + ``point_goal``."""
    bp, br, bv, ba = split_rb(rb)
    rew, raw = im_reward_zero_out_far(rb, ref_now["pos"], ref_now["rot"], ref_now["vel"], ref_now["ang"], point_goal, dof_force, dof_vel,
                                      progress, specs, power_coef, power_reward)
    n = rb.shape[0]
    reset, term = im_reset(torch.zeros(n, dtype=torch.int64), progress, bp[:, reset_body_ids].clone(), ref_now["pos"][:, reset_body_ids].clone(),
                           pass_time, term_dist[..., reset_body_ids])
    if cycle_counter is not None:
        rec = torch.logical_and(~pass_time, cycle_counter > 0)
        reset[rec] = 0
        term[rec] = 0
    tb = track_body_ids
    rp, rr, rv, ra, goal = zero_out_far_refs(obs_v, bp[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb], ref_next["pos"][:, tb],
                                             ref_next["rot"][:, tb], ref_next["vel"][:, tb], ref_next["ang"][:, tb], close_distance, far_distance)
    to = im_obs_variant(obs_v, bp[:, 0], br[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb], rp, rr, rv, ra, 1, upright)
    so = self_obs_smpl_max_general(bp, br, bv, ba, upright=upright)
    return {"obs": torch.cat([so, to], dim=-1), "rew": rew, "raw": raw, "reset": reset, "terminate": term, "point_goal": goal}


# ---------------------------------------------------------------------------------------------------------------------------
# Small training-time options of the step composition (no shipped config enables them; humanoid.py:314-326).
# ---------------------------------------------------------------------------------------------------------------------------
def fut_tracks_dropout(task_obs, uniforms, time_steps, dropout_rate=0.1):
    """_compute_task_obs, humanoid_im.py:804-810: the block of future sample t of env e is zeroed where uniforms[e, t] < 0.1
    (``uniforms`` = the torch.rand(n, T) draw)."""
    n = task_obs.shape[0]
    obs = task_obs.clone().view(n, time_steps, -1)
    mask = uniforms < dropout_rate
    obs[mask, :] = 0
    return obs.view(n, -1)


def add_obs_noise(obs, z):
    """_compute_observations, humanoid_im.py:691-692: obs + randn_like(obs) * 0.1 (``z`` = the draw)."""
    return obs + z * 0.1


def res_action_pd_targets(ref_dof_pos, pd_action_scale, action, dof_pos):
    """HumanoidIm._action_to_pd_targets with res_action, humanoid_im.py:1096-1101."""
    import numpy as np
    pd_tar = ref_dof_pos + pd_action_scale * action
    pd_lower = dof_pos - np.pi / 2
    pd_upper = dof_pos + np.pi / 2
    return torch.maximum(torch.minimum(pd_tar, pd_upper), pd_lower)


def occlude_refs(obs_v, body_pos, body_rot, body_vel, body_ang_vel, ref_pos, ref_rot, ref_vel, ref_ang_vel, occl_idx):
    """The occl_training blocks of _compute_task_obs, humanoid_im.py:778-784 (obs_v 4 / 5 / 6 / 8 / 9: all four quantities) and :827-831 (obs_v 7:
    positions and rotations): an occluded (env, tracked body)'s reference is the simulated state.  occl_idx (N, Jt) bool."""
    ref_pos, ref_rot, ref_vel, ref_ang_vel = ref_pos.clone(), ref_rot.clone(), ref_vel.clone(), ref_ang_vel.clone()
    ref_pos[occl_idx] = body_pos[occl_idx]
    ref_rot[occl_idx] = body_rot[occl_idx]
    if obs_v != 7:
        ref_vel[occl_idx] = body_vel[occl_idx]
        ref_ang_vel[occl_idx] = body_ang_vel[occl_idx]
    return ref_pos, ref_rot, ref_vel, ref_ang_vel
