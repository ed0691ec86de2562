"""Tensor-level wrappers over the C ABI, named after the reference functions they replace.

Every function takes ``torch`` tensors that already live on the GPU, checks
dtype / layout, and enqueues the HIP kernel on torch's CURRENT stream through
``libpulse_hip.so`` (PyTorch is only the allocator / stream provider here).
There is no fallback: CPU tensors are rejected and a missing library raises.

Reference mapping (paths relative to the reference root):
  quat_mul, quat_conjugate                 isaacgym.torch_utils (3P)
  my_quat_rotate ... calc_heading_quat_inv phc/utils/torch_utils.py:45-240
  compute_humanoid_observations_smpl_max   phc/env/tasks/humanoid.py:1675-1731
  compute_imitation_observations_v6 / _v7  phc/env/tasks/humanoid_im.py:1328-1413
  compute_imitation_reward                 phc/env/tasks/humanoid_im.py:1543-1574
  compute_humanoid_im_reset                phc/env/tasks/humanoid_im.py:1600-1628
  discount_values                          phc/learning/common_agent.py:493-505
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS, AmpObsArgs, ImStepArgs, RewardSpecs

DEFAULT_REWARD_SPECS = {"k_pos": 100.0, "k_rot": 10.0, "k_vel": 0.1, "k_ang_vel": 0.1,
                        "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current torch stream of the current device as a hipStream_t.  Called once per kernel launch: the raw accessor is ~20x cheaper than
    building a torch.cuda.Stream object (8 us under a profiler, a tenth of the rollout's host time)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise ValueError(f"{name}: tensor must live on the GPU (pulse_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _c(t, name, dtype=torch.float32):
    t = _dev(t, name, dtype)
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# --------------------------------------------------------------------------- #
# rotation algebra
# --------------------------------------------------------------------------- #
def _rows(t, width, name):
    if t.shape[-1] != width:
        raise ValueError(f"{name}: last dim must be {width}, got {tuple(t.shape)}")
    return t.numel() // width


def quat_mul(a, b):
    a, b = _c(a, "a"), _c(b, "b")
    if a.shape != b.shape:
        raise AssertionError("quat_mul: shape mismatch")  # the reference asserts equal shapes
    out = torch.empty_like(a)
    _lib.check(_lib.load().pulse_quat_mul(_ptr(a), _ptr(b), _ptr(out), _rows(a, 4, "a"), _stream()), "pulse_quat_mul")
    return out


def quat_conjugate(a):
    a = _c(a, "a")
    out = torch.empty_like(a)
    _lib.check(_lib.load().pulse_quat_conjugate(_ptr(a), _ptr(out), _rows(a, 4, "a"), _stream()), "pulse_quat_conjugate")
    return out


def my_quat_rotate(q, v):
    q, v = _c(q, "q"), _c(v, "v")
    m = _rows(q, 4, "q")
    if _rows(v, 3, "v") != m:
        raise ValueError("my_quat_rotate: q and v row counts differ")
    out = torch.empty_like(v)
    _lib.check(_lib.load().pulse_quat_rotate(_ptr(q), _ptr(v), _ptr(out), m, _stream()), "pulse_quat_rotate")
    return out


def quat_to_angle_axis(q):
    q = _c(q, "q")
    m = _rows(q, 4, "q")
    angle = torch.empty(q.shape[:-1], dtype=torch.float32, device=q.device)
    axis = torch.empty(q.shape[:-1] + (3,), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().pulse_quat_to_angle_axis(_ptr(q), _ptr(angle), _ptr(axis), m, _stream()), "pulse_quat_to_angle_axis")
    return angle, axis


def quat_to_exp_map(q):
    q = _c(q, "q")
    out = torch.empty(q.shape[:-1] + (3,), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().pulse_quat_to_exp_map(_ptr(q), _ptr(out), _rows(q, 4, "q"), _stream()), "pulse_quat_to_exp_map")
    return out


def quat_to_tan_norm(q):
    q = _c(q, "q")
    out = torch.empty(q.shape[:-1] + (6,), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().pulse_quat_to_tan_norm(_ptr(q), _ptr(out), _rows(q, 4, "q"), _stream()), "pulse_quat_to_tan_norm")
    return out


def exp_map_to_quat(e):
    e = _c(e, "exp_map")
    out = torch.empty(e.shape[:-1] + (4,), dtype=torch.float32, device=e.device)
    _lib.check(_lib.load().pulse_exp_map_to_quat(_ptr(e), _ptr(out), _rows(e, 3, "exp_map"), _stream()), "pulse_exp_map_to_quat")
    return out


def slerp(q0, q1, t):
    q0, q1 = _c(q0, "q0"), _c(q1, "q1")
    m = _rows(q0, 4, "q0")
    t = _c(t, "t").reshape(-1)
    if q1.shape != q0.shape or t.numel() != m:
        raise ValueError("slerp: shape mismatch")
    out = torch.empty_like(q0)
    _lib.check(_lib.load().pulse_slerp(_ptr(q0), _ptr(q1), _ptr(t), _ptr(out), m, _stream()), "pulse_slerp")
    return out


def calc_heading(q):
    q = _c(q, "q")
    out = torch.empty(q.shape[:-1], dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().pulse_calc_heading(_ptr(q), _ptr(out), _rows(q, 4, "q"), _stream()), "pulse_calc_heading")
    return out


def calc_heading_quat(q):
    q = _c(q, "q")
    out = torch.empty_like(q)
    _lib.check(_lib.load().pulse_calc_heading_quat(_ptr(q), _ptr(out), _rows(q, 4, "q"), 0, _stream()), "pulse_calc_heading_quat")
    return out


def calc_heading_quat_inv(q):
    q = _c(q, "q")
    out = torch.empty_like(q)
    _lib.check(_lib.load().pulse_calc_heading_quat(_ptr(q), _ptr(out), _rows(q, 4, "q"), 1, _stream()), "pulse_calc_heading_quat")
    return out


# --------------------------------------------------------------------------- #
# fused HumanoidIm post-physics step
# --------------------------------------------------------------------------- #
def _specs_struct(specs=None, power_coef=0.0005, power_reward=True):
    s = dict(DEFAULT_REWARD_SPECS)
    if specs:
        s.update(specs)
    return RewardSpecs(s["k_pos"], s["k_rot"], s["k_vel"], s["k_ang_vel"], s["w_pos"], s["w_rot"], s["w_vel"],
                       s["w_ang_vel"], float(power_coef), 1 if power_reward else 0)


def _ids32(ids, device):
    if isinstance(ids, torch.Tensor):
        return ids.to(device=device, dtype=torch.int32).contiguous()
    return torch.tensor(list(ids), dtype=torch.int32, device=device)


def pack_rb(body_pos, body_rot, body_vel, body_ang_vel):
    """Return an (N, J, 13) AoS tensor for the four body tensors.

    When they are the reference-style VIEWS of one Isaac rigid-body buffer
    (phc/env/tasks/humanoid.py:219-222) the buffer itself is returned (zero copy);
    otherwise (gathered copies) the records are re-packed.
    """
    p = _dev(body_pos, "body_pos")
    n, j = p.shape[0], p.shape[1]
    ts = (body_pos, body_rot, body_vel, body_ang_vel)
    offs = (0, 3, 7, 10)
    try:
        same = all(t.untyped_storage().data_ptr() == p.untyped_storage().data_ptr() for t in ts)
    except Exception:
        same = False
    if same and all(t.stride()[-1] == 1 and t.stride()[-2] == 13 for t in ts):
        base_off = p.storage_offset()
        if all(t.storage_offset() - base_off == o for t, o in zip(ts, offs)) and all(t.stride() == p.stride() for t in ts):
            return torch.as_strided(p, (n, j, 13), (p.stride()[0], 13, 1), base_off)
    return torch.cat([_dev(t, "body tensor") for t in ts], dim=-1).contiguous()


# PULSE_IM_DEBUG_POISON_LDS (include/pulse_hip.h): debug builds of a run can ask the fused step to pre-fill its LDS with NaN
_IM_DEBUG_BITS = 0x80000000 if os.environ.get("PULSE_IM_DEBUG_POISON_LDS") == "1" else 0


def _launch_sig(o):
    """Cheap identity of a launch's arguments: device pointers of tensors, values of scalars / id lists, recursively through dicts.  Two calls
    with equal signatures launch the same kernel on the same buffers (a buffer's layout is taken to be fixed by its address and element count)."""
    if o is None:
        return 0
    if isinstance(o, torch.Tensor):
        # (the caching allocator may hand a freed address to a tensor of another size; a view of the same storage may have another dtype / stride)
        return (o.data_ptr(), o.numel(), o.dtype, o.stride())
    if isinstance(o, dict):
        return tuple((k, _launch_sig(v)) for k, v in o.items())
    if isinstance(o, (list, tuple)):
        return tuple(_launch_sig(v) for v in o)
    sig = getattr(o, "launch_signature", None)
    return sig() if sig is not None else o


def _obs_copy_fields(obs_copy, n, cols, dev):
    """(pointer, row stride) of im_step's second observation destination; (None, 0) without one."""
    if obs_copy is None:
        return None, 0
    if (obs_copy.dtype != torch.float32 or obs_copy.device != dev or obs_copy.dim() != 2 or obs_copy.stride(1) != 1 or obs_copy.shape[0] != n or
            obs_copy.shape[1] < cols or obs_copy.stride(0) < cols):
        raise ValueError(f"obs_copy must be a ({n}, >= {cols}) float32 view on {dev} with unit inner stride")
    return obs_copy.data_ptr(), obs_copy.stride(0)


def im_step(rb, *, what, ref_now=None, ref_next=None, time_steps=1, dof_force=None, dof_vel=None,
            progress=None, pass_time=None, cycle_counter=None, track_ids=None, reset_ids=None, term_dist=None,
            reset_use_mean=False, full_body_reward=True, obs_version=6, local_root_obs=True,
            root_height_obs=True, specs=None, power_coef=0.0005, power_reward=True,
            env_ids=None, env_mask=None, obs=None, obs_cols=None, rew=None, rew_raw=None, reset=None,
            terminate=None, clock=None, motion=None, upright=True, enable_early_termination=True, self_obs_version=1,
            force_sensor=None, dof_pos=None, ref_next_dof_pos=None, smpl_params=None, limb_weights=None, recovery_counter=None, cache=None,
            zero_out_far=None, occl_bits=None, occl_reset=False, obs_copy=None):
    """``obs_copy``: a second (N, >= obs_cols) float32 view (any row pitch, e.g. an experience-buffer slot) that receives the same observation rows
    as ``obs``; NOT part of a cached launch's signature -- the pointer is patched into the cached struct on every call.
    ``occl_bits``: (N,) int32, bit j = tracked body j of the env is occluded (occl_training, humanoid_im.py:778-784, 827-831); ``occl_reset``:
    occluded reset bodies never count as fallen (:1178-1183).
    ``zero_out_far``: dict(point_goal (N,) float32 read by the reward stage / written by the task-observation stage, close_distance,
    far_distance) -- the far-masking branch of _compute_task_obs / _compute_reward (humanoid_im.py:763-777, 814-826, 870-887).
    ``cache``: a dict owned by a caller that launches the same step on the same buffers over and over (the env's post-physics step): the
    filled argument struct is kept in it and re-launched as long as the arguments' signature (_launch_sig) does not change -- the struct takes
    ~40 tensor checks to build, a third of the rollout's host time.
    ``clock``: dict(progress_rw, inc, dt, start_times, start_offsets, motion_len, cycle_motion, max_episode_length,
    pass_time_out) -- the episode clock advanced / evaluated in-kernel.  ``motion``: dict(lib, ids, offset, traj_dt, track_rb,
    track_dof_pos, track_dof_vel) -- the reference evaluated in-kernel from a MotionLib instead of ref_now / ref_next.
    Low-level entry: one launch of pulse_im_step.  ``rb`` is (N, J, 13) with unit inner strides
    (the env stride may be larger).  ref_* are dicts with keys pos/rot/vel/ang.  Outputs that are
    not supplied are allocated.  Returns dict(obs, rew, rew_raw, reset, terminate)."""
    lib = _lib.load()
    if cache is not None:
        sig = _launch_sig((rb, what, ref_now, ref_next, time_steps, dof_force, dof_vel, progress, pass_time, cycle_counter, track_ids, reset_ids, term_dist,
                           reset_use_mean, full_body_reward, obs_version, local_root_obs, root_height_obs, specs, power_coef, power_reward, env_ids, env_mask,
                           obs, obs_cols, rew, rew_raw, reset, terminate, clock, motion, upright, enable_early_termination, self_obs_version, force_sensor,
                           dof_pos, ref_next_dof_pos, smpl_params, limb_weights, recovery_counter, zero_out_far, occl_bits, occl_reset, _IM_DEBUG_BITS))
        if cache.get("sig") == sig:
            ca = cache["args"]
            ca.obs_copy, ca.obs_copy_stride = _obs_copy_fields(obs_copy, ca.num_envs, ca.obs_cols, rb.device)
            _lib.check(lib.pulse_im_step(ctypes.byref(ca), _stream()), "pulse_im_step")
            return cache["out"]
        # only launches whose outputs are all caller-owned are replayed (an output this wrapper allocates would be a new buffer every call)
        owned = ((obs is not None or not what & (PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS)) and
                 ((rew is not None and rew_raw is not None) or not what & PULSE_IM_REWARD) and
                 ((reset is not None and terminate is not None) or not what & PULSE_IM_RESET))
    rb = _dev(rb, "rb")
    hist = 1
    if self_obs_version == 2:                      # (N, H, J, 13) history, oldest first (compute_humanoid_observations_smpl_max_v2)
        if rb.dim() != 4 or not rb[0].is_contiguous():
            raise ValueError("self_obs_version 2 takes rb as (N, H, J, 13) with contiguous per-env histories")
        hist = rb.shape[1]
        rb = rb.view(rb.shape[0], hist * rb.shape[2], 13) if rb.is_contiguous() else rb
    if rb.dim() == 4:
        n, j = rb.shape[0], rb.shape[2]
        rb_stride = rb.stride()[0]
    else:
        if rb.dim() != 3 or rb.shape[-1] != 13 or rb.stride()[-1] != 1 or rb.stride()[-2] != 13:
            raise ValueError("rb must be (N, J, 13) with contiguous body records")
        n, j = rb.shape[0], rb.shape[1] // hist
        rb_stride = rb.stride()[0]
    dev = rb.device
    a = ImStepArgs()
    keep = []  # keep temporaries alive until the call returns

    copied = []                                    # arguments that had to be converted into a temporary: such a launch is never replayed from the cache

    def P(t, name, dtype=torch.float32):
        if t is None:
            return None
        t2 = _c(t, name, dtype)
        if t2.data_ptr() != t.data_ptr():
            copied.append(name)
        keep.append(t2)
        return t2.data_ptr()

    a.rb, a.rb_env_stride, a.num_envs, a.num_bodies = rb.data_ptr(), rb_stride, n, j
    a.upright_start, a.enable_early_termination = int(bool(upright)), int(bool(enable_early_termination))
    a.self_obs_version, a.hist_steps = int(self_obs_version), hist
    fsw = 0
    if force_sensor is not None:
        a.force_sensor, a.force_sensor_width = P(force_sensor, "force_sensor"), force_sensor.shape[-1]
        fsw = force_sensor.shape[-1]
    for name, t in (("smpl_params", smpl_params), ("limb_weights", limb_weights)):     # rows appended to the self observation
        if t is not None:
            _dev(t, name)
            if t.dim() != 2 or t.shape[0] != n or t.stride(1) != 1:
                raise ValueError(f"{name}: (num_envs, width) with contiguous rows expected")
            keep.append(t)
            setattr(a, name, t.data_ptr()); setattr(a, name + "_width", t.shape[1]); setattr(a, name + "_stride", t.stride(0))
            fsw += t.shape[1]
    if recovery_counter is not None:
        a.recovery_counter = P(recovery_counter, "recovery_counter", torch.int32)
    if occl_bits is not None:
        if occl_bits.shape != (n,) or occl_bits.dtype != torch.int32:
            raise ValueError("occl_bits: (num_envs,) int32 expected")
        a.occl_bits, a.occl_reset = P(occl_bits, "occl_bits", torch.int32), int(bool(occl_reset))
    if zero_out_far is not None:
        pg = zero_out_far["point_goal"]
        if pg.shape != (n,) or not pg.is_contiguous():
            raise ValueError("zero_out_far.point_goal: contiguous (num_envs,) float32 expected")
        a.zero_out_far, a.point_goal = 1, P(pg, "zero_out_far.point_goal")
        a.close_distance, a.far_distance = float(zero_out_far.get("close_distance", 0.25)), float(zero_out_far.get("far_distance", 3.0))
    a.dof_pos, a.ref_next_dof_pos = P(dof_pos, "dof_pos"), P(ref_next_dof_pos, "ref_next_dof_pos")
    if dof_pos is not None and dof_force is None:
        a.num_dof = dof_pos.shape[-1]
    if env_ids is not None:
        ids_in = env_ids
        env_ids = _c(env_ids, "env_ids", torch.int64)
        if env_ids.data_ptr() != ids_in.data_ptr():
            copied.append("env_ids")
        keep.append(env_ids)
        a.env_ids, a.num_ids = env_ids.data_ptr(), env_ids.numel()
    if env_mask is not None:
        m = env_mask
        if m.dtype == torch.bool:
            m = m.view(torch.uint8)                 # (a view: same memory, in-place edits stay visible to a replayed launch)
        a.env_mask = P(m, "env_mask", torch.uint8)
    if ref_now is not None:
        a.ref_now_pos, a.ref_now_rot = P(ref_now["pos"], "ref_now.pos"), P(ref_now["rot"], "ref_now.rot")
        a.ref_now_vel, a.ref_now_ang = P(ref_now["vel"], "ref_now.vel"), P(ref_now["ang"], "ref_now.ang")
    if ref_next is not None:
        a.ref_next_pos, a.ref_next_vel = P(ref_next["pos"], "ref_next.pos"), P(ref_next["vel"], "ref_next.vel")
        a.ref_next_rot, a.ref_next_ang = P(ref_next.get("rot"), "ref_next.rot"), P(ref_next.get("ang"), "ref_next.ang")
    a.time_steps = time_steps
    a.dof_force, a.dof_vel = P(dof_force, "dof_force"), P(dof_vel, "dof_vel")
    if dof_force is not None:
        a.num_dof = dof_force.shape[-1]
    a.progress = P(progress, "progress", torch.int64)
    if pass_time is not None:
        pt = pass_time.view(torch.uint8) if pass_time.dtype == torch.bool else pass_time
        a.pass_time = P(pt, "pass_time", torch.uint8)
    a.cycle_counter = P(cycle_counter, "cycle_counter", torch.int64)
    # id lists: a Python list is part of the launch signature (its VALUES are), but a tensor that had to be converted (int64 ids, another
    # device) would leave the cached struct pointing at a private copy that later in-place edits of the caller's tensor never reach:
    # such a launch is not replayed from the cache
    if track_ids is not None:
        t = _ids32(track_ids, dev)
        if isinstance(track_ids, torch.Tensor) and t.data_ptr() != track_ids.data_ptr():
            copied.append("track_ids")
        keep.append(t)
        a.track_ids, a.num_track = t.data_ptr(), t.numel()
    if reset_ids is not None:
        t = _ids32(reset_ids, dev)
        if isinstance(reset_ids, torch.Tensor) and t.data_ptr() != reset_ids.data_ptr():
            copied.append("reset_ids")
        keep.append(t)
        a.reset_ids, a.num_reset = t.data_ptr(), t.numel()
    a.term_dist = P(term_dist, "term_dist")
    a.reset_use_mean, a.full_body_reward = int(reset_use_mean), int(full_body_reward)
    a.what, a.obs_version = what | _IM_DEBUG_BITS, obs_version
    a.local_root_obs, a.root_height_obs = int(local_root_obs), int(root_height_obs)
    a.specs = _specs_struct(specs, power_coef, power_reward)

    out = {}
    if what & (PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS):
        sw = lib.pulse_self_obs_width_ex(j, int(root_height_obs), int(self_obs_version), hist, fsw)
        tw = lib.pulse_task_obs_width(obs_version, a.num_track, time_steps) if what & PULSE_IM_TASK_OBS else 0
        width = sw + tw
        if obs is None:
            obs = torch.empty(n, width if obs_cols is None else obs_cols, dtype=torch.float32, device=dev)
        _dev(obs, "obs")
        if obs.stride()[-1] != 1:
            raise ValueError("obs rows must be contiguous")
        a.obs, a.obs_stride = obs.data_ptr(), obs.stride()[0]
        a.obs_cols = width if obs_cols is None else obs_cols
        out["obs"] = obs
        a.obs_copy, a.obs_copy_stride = _obs_copy_fields(obs_copy, n, a.obs_cols, dev)
    elif obs_copy is not None:
        raise ValueError("obs_copy without an observation stage")
    if what & PULSE_IM_REWARD:
        rw = 5 if power_reward else 4
        rew = torch.empty(n, dtype=torch.float32, device=dev) if rew is None else _dev(rew, "rew")
        rew_raw = torch.empty(n, rw, dtype=torch.float32, device=dev) if rew_raw is None else _dev(rew_raw, "rew_raw")
        if not (rew.is_contiguous() and rew_raw.is_contiguous() and rew_raw.shape[-1] == rw):
            raise ValueError("rew / rew_raw must be contiguous, rew_raw (N, 4|5)")
        a.rew, a.rew_raw = rew.data_ptr(), rew_raw.data_ptr()
        out["rew"], out["rew_raw"] = rew, rew_raw
    if what & PULSE_IM_RESET:
        reset = torch.empty(n, dtype=torch.int64, device=dev) if reset is None else _dev(reset, "reset", torch.int64)
        terminate = torch.empty(n, dtype=torch.int64, device=dev) if terminate is None else _dev(terminate, "terminate", torch.int64)
        a.reset, a.terminate = reset.data_ptr(), terminate.data_ptr()
        out["reset"], out["terminate"] = reset, terminate
    if clock is not None:
        a.progress_rw, a.progress_inc = P(clock.get("progress_rw"), "clock.progress_rw", torch.int64), int(clock.get("inc", 0))
        a.clock_dt = float(clock.get("dt", 0.0))
        a.clock_start_times, a.clock_start_offsets = P(clock.get("start_times"), "clock.start_times"), P(clock.get("start_offsets"), "clock.start_offsets")
        a.clock_motion_len = P(clock.get("motion_len"), "clock.motion_len")
        a.cycle_motion, a.max_episode_length = int(bool(clock.get("cycle_motion", False))), int(clock.get("max_episode_length", 0))
        pto = clock.get("pass_time_out")
        if pto is not None:
            a.pass_time_out = P(pto.view(torch.uint8) if pto.dtype == torch.bool else pto, "clock.pass_time_out", torch.uint8)
    if motion is not None:
        a.use_motion = 1
        motion["lib"].fill_tables(a.motion)
        a.motion_ids, a.motion_offset = P(motion["ids"], "motion.ids", torch.int64), P(motion.get("offset"), "motion.offset")
        a.traj_dt = float(motion.get("traj_dt", 0.0))
        trb = motion.get("track_rb")
        if trb is not None:
            a.track_rb, a.track_rb_stride = P(trb, "motion.track_rb"), trb.stride(0)
        a.track_dof_pos, a.track_dof_vel = P(motion.get("track_dof_pos"), "motion.track_dof_pos"), P(motion.get("track_dof_vel"), "motion.track_dof_vel")
    _lib.check(lib.pulse_im_step(ctypes.byref(a), _stream()), "pulse_im_step")
    if cache is not None:
        cache.clear()
        if owned and not copied:
            cache.update(sig=sig, args=a, keep=keep + [rb], out=out)
    return out


def compute_humanoid_observations_smpl_max(body_pos, body_rot, body_vel, body_ang_vel, smpl_params=None,
                                           limb_weight_params=None, local_root_obs=True, root_height_obs=True,
                                           upright=True, has_smpl_params=False, has_limb_weight_params=False):
    """phc/env/tasks/humanoid.py:1675-1731 (shape / limb-weight rows appended when has_* is set, :1724-1728); upright=False applies
    remove_base_rot (:1616-1620)."""
    rb = pack_rb(body_pos, body_rot, body_vel, body_ang_vel)
    return im_step(rb, what=PULSE_IM_SELF_OBS, local_root_obs=local_root_obs, root_height_obs=root_height_obs, upright=upright,
                   smpl_params=smpl_params if has_smpl_params else None, limb_weights=limb_weight_params if has_limb_weight_params else None)["obs"]


def compute_humanoid_observations_smpl_max_v2(body_pos, body_rot, body_vel, body_ang_vel, smpl_params=None, limb_weight_params=None,
                                              local_root_obs=True, root_height_obs=True, upright=True, has_smpl_params=False,
                                              has_limb_weight_params=False, time_steps=1):
    """phc/env/tasks/humanoid.py:1734-1786: (B, T, J, .) history in the heading frame of the newest root."""
    if has_smpl_params or has_limb_weight_params or not local_root_obs:
        raise NotImplementedError("the reference itself raises for these options (humanoid.py:1766-1783)")
    rb = torch.cat([body_pos, body_rot, body_vel, body_ang_vel], dim=-1).contiguous()
    return im_step(rb, what=PULSE_IM_SELF_OBS, local_root_obs=True, root_height_obs=root_height_obs, upright=upright, self_obs_version=2)["obs"]


def compute_humanoid_observations_smpl_max_v3(body_pos, body_rot, body_vel, body_ang_vel, force_sensor_readings, smpl_params=None,
                                              limb_weight_params=None, local_root_obs=True, root_height_obs=True, upright=True,
                                              has_smpl_params=False, has_limb_weight_params=False):
    """phc/env/tasks/humanoid.py:1789-1849: _smpl_max + the force-sensor readings (+ shape / limb-weight rows, :1843-1847)."""
    rb = pack_rb(body_pos, body_rot, body_vel, body_ang_vel)
    return im_step(rb, what=PULSE_IM_SELF_OBS, local_root_obs=local_root_obs, root_height_obs=root_height_obs, upright=upright,
                   self_obs_version=3, force_sensor=force_sensor_readings,
                   smpl_params=smpl_params if has_smpl_params else None, limb_weights=limb_weight_params if has_limb_weight_params else None)["obs"]


def remove_base_rot(quat):
    """phc/env/tasks/humanoid.py:1616-1620."""
    base = torch.tensor([[-0.5, -0.5, -0.5, 0.5]], dtype=torch.float32, device=quat.device).repeat(quat.shape[0], 1)
    return quat_mul(quat, base)


def _split_task_only(full, self_w):
    return full[:, self_w:]


def compute_imitation_observations_v6(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                                      ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, time_steps, upright=True):
    """phc/env/tasks/humanoid_im.py:1328-1378.  body_* are the tracked subset (B, Jt, .); the root
    is passed separately exactly as in the reference."""
    return _task_obs(6, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                     ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, time_steps, upright)


def compute_imitation_observations(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                                   ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, time_steps, upright=True):
    """obs_v 1, phc/env/tasks/humanoid_im.py:1222-1256."""
    return _task_obs(1, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_body_pos, ref_body_rot, ref_body_vel,
                     ref_body_ang_vel, time_steps, upright)


def compute_imitation_observations_v2(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, dof_pos, ref_body_pos, ref_body_rot,
                                      ref_body_vel, ref_body_ang_vel, ref_dof_pos, time_steps, upright=True):
    """obs_v 2, phc/env/tasks/humanoid_im.py:1259-1297: v1 + dof differences (dof_pos / ref_dof_pos: (B, Jt - 1, 3) subsets)."""
    return _task_obs(2, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_body_pos, ref_body_rot, ref_body_vel,
                     ref_body_ang_vel, time_steps, upright, dof_pos=dof_pos, ref_dof_pos=ref_dof_pos)


def compute_imitation_observations_v3(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                                      ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, time_steps, upright=True):
    """obs_v 3, phc/env/tasks/humanoid_im.py:1300-1325."""
    return _task_obs(3, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_body_pos, ref_body_rot, ref_body_vel,
                     ref_body_ang_vel, time_steps, upright)


def compute_imitation_observations_v8(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                                      ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, time_steps, upright=True):
    """obs_v 8, phc/env/tasks/humanoid_im.py:1415-1481 (time_steps 1)."""
    return _task_obs(8, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_body_pos, ref_body_rot, ref_body_vel,
                     ref_body_ang_vel, time_steps, upright)


def compute_imitation_observations_v9(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                                      ref_body_pos, ref_body_rot, ref_root_vel, ref_root_ang_vel, time_steps, upright=True):
    """obs_v 9, phc/env/tasks/humanoid_im.py:1484-1540: the reference passes only the ROOT reference velocities."""
    b, jt = body_pos.shape[0], body_pos.shape[1]
    rv = torch.zeros(b * time_steps, jt, 3, dtype=torch.float32, device=body_pos.device)
    ra = torch.zeros_like(rv)
    rv[:, 0], ra[:, 0] = ref_root_vel.reshape(-1, 3), ref_root_ang_vel.reshape(-1, 3)
    return _task_obs(9, root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel, ref_body_pos, ref_body_rot, rv, ra, time_steps, upright)


def compute_imitation_observations_v7(root_pos, root_rot, body_pos, body_vel, ref_body_pos, ref_body_vel, time_steps, upright=True):
    """phc/env/tasks/humanoid_im.py:1381-1413."""
    z4 = torch.zeros(body_pos.shape[:-1] + (4,), dtype=torch.float32, device=body_pos.device)
    z4[..., 3] = 1.0
    z3 = torch.zeros_like(body_pos)
    return _task_obs(7, root_pos, root_rot, body_pos, z4, body_vel, z3, ref_body_pos, None, ref_body_vel, None, time_steps, upright)


def _task_obs(version, root_pos, root_rot, bp, br, bv, ba, rp, rr, rv, ra, time_steps, upright=True, dof_pos=None, ref_dof_pos=None):
    # The kernel indexes bodies through a track list and reads the root from body 0, so build
    # a (B, 1 + Jt, 13) record array: slot 0 = root, slots 1.. = the tracked subset.
    b, jt = bp.shape[0], bp.shape[1]
    dev = bp.device
    root = torch.zeros(b, 1, 13, dtype=torch.float32, device=dev)
    root[:, 0, 0:3] = root_pos
    root[:, 0, 3:7] = root_rot
    rb = torch.cat([root, torch.cat([bp, br, bv, ba], dim=-1)], dim=1).contiguous()
    j = 1 + jt

    def pad(x, w):  # (B*T, Jt, w) -> (B*T, 1+Jt, w)
        x = x.reshape(b * time_steps, jt, w)
        return torch.cat([torch.zeros(b * time_steps, 1, w, dtype=torch.float32, device=dev), x], dim=1).contiguous()

    ref_next = {"pos": pad(rp, 3), "vel": pad(rv, 3)}
    if version != 7:
        ref_next["rot"], ref_next["ang"] = pad(rr, 4), pad(ra, 3)
    lib = _lib.load()
    sw = lib.pulse_self_obs_width(j, 1)
    tw = lib.pulse_task_obs_width(version, jt, time_steps)
    full = torch.empty(b, sw + tw, dtype=torch.float32, device=dev)
    kw = {}
    if version == 2:
        # joint (track id - 1) of the packed array = subset joint of tracked body i >= 1 (the root carries no dof)
        def dof_full(d):
            f = torch.zeros(b, 3 * (j - 1), dtype=torch.float32, device=dev)
            f[:, 3:] = d.reshape(b, -1)
            return f
        kw = {"dof_pos": dof_full(dof_pos), "ref_next_dof_pos": dof_full(ref_dof_pos)}
    im_step(rb, what=PULSE_IM_TASK_OBS, ref_next=ref_next, time_steps=time_steps, obs_version=version,
            track_ids=list(range(1, j)), obs=full, upright=upright, **kw)
    return full[:, sw:]


def compute_imitation_reward(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                             ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, rwd_specs):
    """phc/env/tasks/humanoid_im.py:1543-1574 -> (reward (N,), reward_raw (N,4))."""
    rb = pack_rb(body_pos, body_rot, body_vel, body_ang_vel)
    n = rb.shape[0]
    dev = rb.device
    ref = {"pos": ref_body_pos, "rot": ref_body_rot, "vel": ref_body_vel, "ang": ref_body_ang_vel}
    out = im_step(rb, what=PULSE_IM_REWARD, ref_now=ref, specs=rwd_specs, power_reward=False,
                  progress=torch.zeros(n, dtype=torch.int64, device=dev))
    return out["rew"], out["rew_raw"]


def compute_humanoid_im_reset(reset_buf, progress_buf, contact_buf, contact_body_ids, rigid_body_pos, ref_body_pos,
                              pass_time, enable_early_termination, termination_distance, disableCollision, use_mean):
    """phc/env/tasks/humanoid_im.py:1600-1628.  rigid_body_pos / ref_body_pos are already restricted
    to the reset bodies (N, Jr, 3); termination_distance is (Jr,) or (1, Jr)."""
    n, jr = rigid_body_pos.shape[0], rigid_body_pos.shape[1]
    dev = rigid_body_pos.device
    if (not enable_early_termination) or disableCollision:
        term = torch.zeros_like(reset_buf)
        return torch.where(pass_time, torch.ones_like(reset_buf), term), term
    rb = torch.zeros(n, jr, 13, dtype=torch.float32, device=dev)
    rb[..., 0:3] = rigid_body_pos
    rb[..., 6] = 1.0
    z3 = torch.zeros(n, jr, 3, dtype=torch.float32, device=dev)
    z4 = torch.zeros(n, jr, 4, dtype=torch.float32, device=dev)
    ref = {"pos": ref_body_pos, "rot": z4, "vel": z3, "ang": z3}
    td = termination_distance.reshape(-1).to(torch.float32)
    if td.numel() == 1:
        td = td.expand(jr)
    out = im_step(rb, what=PULSE_IM_RESET, ref_now=ref, progress=progress_buf, pass_time=pass_time,
                  reset_ids=list(range(jr)), term_dist=td.contiguous(), reset_use_mean=use_mean)
    return out["reset"], out["terminate"]


# --------------------------------------------------------------------------- #
# downstream tasks (speed / reach / strike): one launch of pulse_task_step
# --------------------------------------------------------------------------- #
def task_step(task, rb, *, what, prev_root_pos=None, dt=1.0 / 30.0, tar_speed=None, tar_pos=None, reach_body_id=0, tar_states=None,
              tar_contact_forces=None, strike_body_ids=None, contact_forces=None, contact_body_ids=None, termination_heights=None,
              progress=None, max_episode_length=300.0, enable_early_termination=True, dof_force=None, dof_vel=None, power_coef=0.0005,
              power_reward=False, obs=None, obs_offset=0, rew=None, rew_raw=None, reset=None, terminate=None, env_ids=None, env_mask=None):
    """``task``: 'speed' | 'reach' | 'strike'.  ``rb`` (N, J, 13).  Returns dict(obs, rew, rew_raw, reset, terminate) of what was asked."""
    from ._lib import TASK_OBS, TASK_REACH, TASK_RESET, TASK_REWARD, TASK_SPEED, TASK_STRIKE, TaskStepArgs
    lib = _lib.load()
    rb = _dev(rb, "rb")
    n, j = rb.shape[0], rb.shape[1]
    dev = rb.device
    a = TaskStepArgs()
    keep = []

    def P(t, name, dtype=torch.float32):
        if t is None:
            return None
        t = _c(t, name, dtype)
        keep.append(t)
        return t.data_ptr()

    a.task = {"speed": TASK_SPEED, "reach": TASK_REACH, "strike": TASK_STRIKE}[task]
    a.what, a.num_envs = what, n
    a.rb, a.rb_env_stride, a.num_bodies = rb.data_ptr(), rb.stride()[0], j
    if env_ids is not None:
        env_ids = _c(env_ids, "env_ids", torch.int64)
        keep.append(env_ids)
        a.env_ids, a.num_ids = env_ids.data_ptr(), env_ids.numel()
    if env_mask is not None:
        a.env_mask = P(env_mask.view(torch.uint8) if env_mask.dtype == torch.bool else env_mask, "env_mask", torch.uint8)
    a.prev_root_pos, a.dt = P(prev_root_pos, "prev_root_pos"), float(dt)
    a.tar_speed, a.tar_pos, a.reach_body_id = P(tar_speed, "tar_speed"), P(tar_pos, "tar_pos"), int(reach_body_id)
    a.tar_states, a.tar_contact_forces = P(tar_states, "tar_states"), P(tar_contact_forces, "tar_contact_forces")
    for name, ids in (("strike", strike_body_ids), ("contact", contact_body_ids)):
        if ids is not None:
            t = _ids32(ids, dev)
            keep.append(t)
            if name == "strike":
                a.strike_body_ids, a.num_strike = t.data_ptr(), t.numel()
            else:
                a.contact_body_ids, a.num_contact_ids = t.data_ptr(), t.numel()
    a.contact_forces, a.termination_heights = P(contact_forces, "contact_forces"), P(termination_heights, "termination_heights")
    a.progress, a.max_episode_length = P(progress, "progress", torch.int64), float(max_episode_length)
    a.enable_early_termination = int(bool(enable_early_termination))
    a.dof_force, a.dof_vel = P(dof_force, "dof_force"), P(dof_vel, "dof_vel")
    a.num_dof = dof_force.shape[-1] if dof_force is not None else 0
    a.power_coef, a.power_reward = float(power_coef), int(bool(power_reward))
    out = {}
    if what & TASK_OBS:
        w = lib.pulse_task_obs_size(a.task)
        if obs is None:
            obs = torch.empty(n, obs_offset + w, dtype=torch.float32, device=dev)
        _dev(obs, "obs")
        a.obs, a.obs_stride, a.obs_offset = obs.data_ptr(), obs.stride()[0], int(obs_offset)
        out["obs"] = obs
    if what & TASK_REWARD:
        rw = 2 if power_reward else 1
        rew = torch.empty(n, dtype=torch.float32, device=dev) if rew is None else _dev(rew, "rew")
        rew_raw = torch.empty(n, rw, dtype=torch.float32, device=dev) if rew_raw is None else _dev(rew_raw, "rew_raw")
        a.rew, a.rew_raw, a.rew_raw_width = rew.data_ptr(), rew_raw.data_ptr(), rew_raw.shape[-1]
        out["rew"], out["rew_raw"] = rew, rew_raw
    if what & TASK_RESET:
        reset = torch.empty(n, dtype=torch.int64, device=dev) if reset is None else _dev(reset, "reset", torch.int64)
        terminate = torch.empty(n, dtype=torch.int64, device=dev) if terminate is None else _dev(terminate, "terminate", torch.int64)
        a.reset, a.terminate = reset.data_ptr(), terminate.data_ptr()
        out["reset"], out["terminate"] = reset, terminate
    _lib.check(lib.pulse_task_step(ctypes.byref(a), _stream()), "pulse_task_step")
    return out


# --------------------------------------------------------------------------- #
# AMP observation
# --------------------------------------------------------------------------- #
def amp_obs_width(num_joints, num_key_bodies, root_height_obs=True):
    return _lib.load().pulse_amp_obs_width(num_joints, num_key_bodies, int(root_height_obs))


def build_amp_observations_smpl(rb, dof_pos, dof_vel, key_body_ids, *, joint_ids=None, zero_joints=(), local_root_obs=True,
                                root_height_obs=True, out=None, env_ids=None, env_mask=None, hist_steps=0, window_out=None):
    """phc/env/tasks/humanoid_amp.py:925-969 on the (N, bodies, 13) rigid-body records (root = body 0) and the
    (N, num_dof) dof tensors.  ``joint_ids`` = dof_subset expressed in joints; ``zero_joints`` = joints whose dofs
    read as zero (:636-639).  Writes the first W columns of ``out`` rows (any row pitch) and returns ``out``."""
    lib = _lib.load()
    rb = _dev(rb, "rb")
    n = rb.shape[0]
    dev = rb.device
    dof_pos, dof_vel = _c(dof_pos, "dof_pos"), _c(dof_vel, "dof_vel")
    kb = _ids32(key_body_ids, dev)
    ji = _ids32(joint_ids, dev) if joint_ids is not None else None
    nj = ji.numel() if ji is not None else dof_pos.shape[-1] // 3
    w = lib.pulse_amp_obs_width(nj, kb.numel(), int(root_height_obs))
    if out is None:
        out = torch.empty(n, w, dtype=torch.float32, device=dev)
    a = AmpObsArgs()
    a.rb, a.rb_env_stride = rb.data_ptr(), rb.stride()[0]
    a.dof_pos, a.dof_vel, a.num_dof, a.num_envs = dof_pos.data_ptr(), dof_vel.data_ptr(), dof_pos.shape[-1], n
    keep = [kb, ji, dof_pos, dof_vel]
    if env_ids is not None:
        env_ids = _c(env_ids, "env_ids", torch.int64)
        keep.append(env_ids)
        a.env_ids, a.num_ids = env_ids.data_ptr(), env_ids.numel()
    if env_mask is not None:
        m = env_mask.view(torch.uint8) if env_mask.dtype == torch.bool else env_mask
        m = _c(m, "env_mask", torch.uint8)
        keep.append(m)
        a.env_mask = m.data_ptr()
    a.joint_ids, a.num_joints = (ji.data_ptr() if ji is not None else None), nj
    a.zero_joint_mask = sum(1 << int(j) for j in zero_joints)
    a.key_body_ids, a.num_key_bodies = kb.data_ptr(), kb.numel()
    a.local_root_obs, a.root_height_obs = int(local_root_obs), int(root_height_obs)
    a.out, a.out_stride = out.data_ptr(), out.stride()[0]
    if hist_steps and hist_steps > 1:
        # ``out`` is slot 0 of the (N, hist_steps, W) history: shift + current frame (+ copy of the finished window) in this launch
        a.hist_steps = int(hist_steps)
        if window_out is not None:
            if window_out.dtype != torch.float32 or not window_out.is_cuda or window_out.stride(-1) != 1 or window_out.shape[0] != n:
                raise TypeError("build_amp_observations_smpl: window_out must be a float32 CUDA tensor with one row per env")
            a.window_out, a.window_stride = window_out.data_ptr(), window_out.stride(0)
    elif window_out is not None:
        raise ValueError("build_amp_observations_smpl: window_out goes with hist_steps > 1")
    _lib.check(lib.pulse_amp_obs(ctypes.byref(a), _stream()), "pulse_amp_obs")
    return out


def amp_hist_init(motion_lib, motion_ids, start_times, dt, env_mask, hist, key_body_ids, *, joint_ids=None, local_root_obs=True, root_height_obs=True):
    """_init_amp_obs_ref (humanoid_amp.py:531-563) for the masked envs: slots 1 .. S-1 of ``hist`` (N, S, W) := the AMP frames of each env's
    motion at start_times - dt * (k + 1), in one launch (pulse_amp_hist_init)."""
    lib = _lib.load()
    dev = hist.device
    if hist.dtype != torch.float32 or hist.dim() != 3 or hist.stride(2) != 1 or not hist.is_cuda:
        raise TypeError("amp_hist_init: hist must be a (N, S, W) float32 CUDA tensor")
    n, s_, w = hist.shape
    ids = _c(motion_ids, "motion_ids", torch.int64)
    st = _c(start_times, "start_times", torch.float32)
    if ids.numel() != n or st.numel() != n:
        raise ValueError("amp_hist_init: one motion id / start time per env")
    m = env_mask.view(torch.uint8) if env_mask.dtype == torch.bool else env_mask
    m = _c(m, "env_mask", torch.uint8)
    kb = _ids32(key_body_ids, dev)
    ji = _ids32(joint_ids, dev) if joint_ids is not None else None
    nj = ji.numel() if ji is not None else motion_lib.num_bodies - 1
    if lib.pulse_amp_obs_width(nj, kb.numel(), int(root_height_obs)) != w:
        raise ValueError("amp_hist_init: the history's frame width does not match the joint / key-body selection")
    a = _lib.AmpHistArgs()
    motion_lib.fill_tables(a.tab)
    a.motion_ids, a.start_times, a.dt = ids.data_ptr(), st.data_ptr(), float(dt)
    a.num_envs, a.env_mask, a.hist_steps = n, m.data_ptr(), s_
    a.joint_ids, a.num_joints = (ji.data_ptr() if ji is not None else None), nj
    a.key_body_ids, a.num_key_bodies = kb.data_ptr(), kb.numel()
    a.local_root_obs, a.root_height_obs = int(local_root_obs), int(root_height_obs)
    a.hist, a.env_stride, a.step_stride = hist.data_ptr(), hist.stride(0), hist.stride(1)
    _lib.check(lib.pulse_amp_hist_init(ctypes.byref(a), _stream()), "pulse_amp_hist_init")
    return hist


# --------------------------------------------------------------------------- #
# GAE
# --------------------------------------------------------------------------- #
def discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values, gamma, tau, return_returns=False):
    """CommonAgent.discount_values, phc/learning/common_agent.py:493-505.

    Tensors are (T, N, 1) [dones (T, N), uint8 or float] in ANY strides that all agree
    (the reference's time-major layout or this framework's env-major buffers viewed time-major).
    """
    r = _dev(mb_rewards, "mb_rewards")
    t, n = r.shape[0], r.shape[1]
    v, nv = _dev(mb_values, "mb_values"), _dev(mb_next_values, "mb_next_values")
    d = mb_fdones
    if d.dtype != torch.uint8:
        d = (d != 0).to(torch.uint8)
    _dev(d, "mb_fdones", torch.uint8)
    st, sn = r.stride()[0], r.stride()[1]
    for x, name in ((v, "values"), (nv, "next_values")):
        if (x.stride()[0], x.stride()[1]) != (st, sn) or x.shape[:2] != r.shape[:2]:
            raise ValueError(f"discount_values: {name} layout differs from rewards")
    if (d.stride()[0], d.stride()[1]) != (st, sn):
        # bring dones to the rewards' layout (uint8, tiny)
        d2 = torch.empty_strided((t, n), (st, sn), dtype=torch.uint8, device=r.device)
        d2.copy_(d.reshape(t, n))
        d = d2
    advs = torch.empty_strided(r.shape, r.stride(), dtype=torch.float32, device=r.device)
    rets = torch.empty_strided(r.shape, r.stride(), dtype=torch.float32, device=r.device) if return_returns else None
    gt = float(gamma) * float(tau)  # Python evaluates gamma * tau in double first
    _lib.check(_lib.load().pulse_gae(_ptr(r), _ptr(v), _ptr(nv), _ptr(d), t, n, st, sn, float(gamma), gt,
                                     _ptr(advs), _ptr(rets), _stream()), "pulse_gae")
    return (advs, rets) if return_returns else advs


# --------------------------------------------------------------------------- #
# trajectory following over a height field (HumanoidTraj / HumanoidPedestrianTerrain)
# --------------------------------------------------------------------------- #
def traj_generate(rb, verts, u_dtheta, u_sharp, sharp_mask, u_heading, u_dspeed, u_speed0, *, episode_dur, dtheta_max=2.0, speed_min=0.0,
                  speed_max=3.0, accel_max=2.0, env_mask=None):
    """TrajGenerator.reset (phc/utils/traj_generator.py:60-123) for the masked envs from uniform draws in the reference's order.
    ``verts`` (N, num_verts, 3) is updated in place; trajectories start at each env's root xy."""
    from ._lib import TrajGenArgs
    rb, verts = _dev(rb, "rb"), _dev(verts, "verts")
    n, nv = verts.shape[0], verts.shape[1]
    if not verts.is_contiguous() or verts.shape[2] != 3:
        raise ValueError("verts: contiguous (N, num_verts, 3) expected")
    seg_dt = episode_dur / (nv - 1)
    a = TrajGenArgs()
    keep = []

    def P(t, name, shape, dtype=torch.float32):
        t = _c(t, name, dtype)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        keep.append(t)
        return t.data_ptr()
    a.num_envs, a.num_verts = n, nv
    if env_mask is not None:
        a.env_mask = P(env_mask.view(torch.uint8) if env_mask.dtype == torch.bool else env_mask, "env_mask", (n,), torch.uint8)
    a.rb, a.rb_env_stride = rb.data_ptr(), rb.stride(0)
    a.u_dtheta, a.u_sharp, a.u_dspeed = P(u_dtheta, "u_dtheta", (n, nv - 1)), P(u_sharp, "u_sharp", (n, nv - 1)), P(u_dspeed, "u_dspeed", (n, nv - 1))
    a.sharp_mask = P(sharp_mask.view(torch.uint8) if sharp_mask.dtype == torch.bool else sharp_mask, "sharp_mask", (n, nv - 1), torch.uint8)
    a.u_heading, a.u_speed0 = P(u_heading, "u_heading", (n,)), P(u_speed0, "u_speed0", (n,))
    a.dtheta_scale, a.dspeed_scale, a.seg_dt = float(dtheta_max * seg_dt), float(accel_max * seg_dt), float(seg_dt)
    a.speed_min, a.speed_max = float(speed_min), float(speed_max)
    a.verts = verts.data_ptr()
    _lib.check(_lib.load().pulse_traj_generate(ctypes.byref(a), _stream()), "pulse_traj_generate")
    return verts


def traj_step(rb, verts, progress, *, what, dt, episode_dur, num_samples=10, sample_timestep=0.5, upright=True, heightsamples=None,
              horizontal_scale=0.1, vertical_scale=0.005, height_points=None, sensor_body=0, center_points=None, use_center_height=True,
              height_meas_scale=5.0, dof_force=None, dof_vel=None, power_coef=0.0005, power_reward=False, fuzzy_target=False, contact_forces=None,
              contact_body_ids=None, termination_heights=None, max_episode_length=300.0, fail_dist=4.0, enable_early_termination=True,
              terrain_reset=True, disable_collision=False, obs=None, obs_offset=0, rew=None, rew_raw=None, reset=None, terminate=None,
              env_ids=None, env_mask=None):
    """One launch of pulse_traj_step (include/pulse_hip.h 2a''): task observation [10 trajectory samples x 2 | height map], location (+ power)
    reward, reset / terminate.  ``height_points`` None = no terrain observation (HumanoidTraj); ``heightsamples`` None with height points =
    terrainType 'plane'.  Returns dict(obs, rew, rew_raw, reset, terminate) of what was asked."""
    from ._lib import TASK_OBS, TASK_RESET, TASK_REWARD, TrajStepArgs
    rb = _dev(rb, "rb")
    n, j = rb.shape[0], rb.shape[1]
    dev = rb.device
    a = TrajStepArgs()
    keep = []

    def P(t, name, dtype=torch.float32):
        if t is None:
            return None
        t = _c(t, name, dtype)
        keep.append(t)
        return t.data_ptr()
    a.what, a.num_envs = what, n
    if env_ids is not None:
        env_ids = _c(env_ids, "env_ids", torch.int64)
        keep.append(env_ids)
        a.env_ids, a.num_ids = env_ids.data_ptr(), env_ids.numel()
    if env_mask is not None:
        a.env_mask = P(env_mask.view(torch.uint8) if env_mask.dtype == torch.bool else env_mask, "env_mask", torch.uint8)
    a.rb, a.rb_env_stride, a.num_bodies, a.upright_start = rb.data_ptr(), rb.stride(0), j, int(bool(upright))
    a.progress, a.dt = P(progress, "progress", torch.int64), float(dt)
    verts = _dev(verts, "verts")
    if verts.shape[0] != n or verts.dim() != 3 or verts.shape[2] != 3 or not verts.is_contiguous():
        raise ValueError("verts: contiguous (N, num_verts, 3) expected")
    nv = verts.shape[1]
    a.verts, a.num_verts, a.traj_dur = verts.data_ptr(), nv, float(nv * (episode_dur / (nv - 1)))
    a.num_samples, a.sample_timestep = int(num_samples), float(sample_timestep)
    nh = 0
    if height_points is not None:
        nh = height_points.shape[0]
        a.height_points, a.num_height_points, a.sensor_body = P(height_points, "height_points"), nh, int(sensor_body)
        if heightsamples is not None:
            if heightsamples.dtype != torch.int16 or heightsamples.dim() != 2 or not heightsamples.is_contiguous() or not heightsamples.is_cuda:
                raise TypeError("heightsamples: contiguous (rows, cols) int16 CUDA tensor expected")
            keep.append(heightsamples)
            a.heightsamples, a.map_rows, a.map_cols = heightsamples.data_ptr(), heightsamples.shape[0], heightsamples.shape[1]
        a.horizontal_scale, a.vertical_scale = float(horizontal_scale), float(vertical_scale)
        if center_points is not None:
            a.center_points, a.num_center_points = P(center_points, "center_points"), center_points.shape[0]
        a.use_center_height, a.height_meas_scale = int(bool(use_center_height)), float(height_meas_scale)
    a.dof_force, a.dof_vel = P(dof_force, "dof_force"), P(dof_vel, "dof_vel")
    a.num_dof = dof_force.shape[-1] if dof_force is not None else 0
    a.power_coef, a.power_reward, a.fuzzy_target = float(power_coef), int(bool(power_reward)), int(bool(fuzzy_target))
    if contact_body_ids is not None:
        t = _ids32(contact_body_ids, dev)
        keep.append(t)
        a.contact_body_ids, a.num_contact_ids = t.data_ptr(), t.numel()
    a.contact_forces, a.termination_heights = P(contact_forces, "contact_forces"), P(termination_heights, "termination_heights")
    a.max_episode_length, a.fail_dist = float(max_episode_length), float(fail_dist)
    a.enable_early_termination, a.terrain_reset, a.disable_collision = int(bool(enable_early_termination)), int(bool(terrain_reset)), int(bool(disable_collision))
    out = {}
    if what & TASK_OBS:
        w = 2 * int(num_samples) + nh
        if obs is None:
            obs = torch.zeros(n, (obs_offset + w + 3) // 4 * 4, dtype=torch.float32, device=dev)
        _dev(obs, "obs")
        a.obs, a.obs_stride, a.obs_offset = obs.data_ptr(), obs.stride(0), int(obs_offset)
        out["obs"] = obs
    if what & TASK_REWARD:
        rew = torch.empty(n, dtype=torch.float32, device=dev) if rew is None else _dev(rew, "rew")
        rew_raw = torch.empty(n, 2, dtype=torch.float32, device=dev) if rew_raw is None else _dev(rew_raw, "rew_raw")
        if not rew_raw.is_contiguous() or rew_raw.shape[-1] != 2:
            raise ValueError("rew_raw: contiguous (N, 2) expected")
        a.rew, a.rew_raw = rew.data_ptr(), rew_raw.data_ptr()
        out["rew"], out["rew_raw"] = rew, rew_raw
    if what & TASK_RESET:
        reset = torch.empty(n, dtype=torch.int64, device=dev) if reset is None else _dev(reset, "reset", torch.int64)
        terminate = torch.empty(n, dtype=torch.int64, device=dev) if terminate is None else _dev(terminate, "terminate", torch.int64)
        a.reset, a.terminate = reset.data_ptr(), terminate.data_ptr()
        out["reset"], out["terminate"] = reset, terminate
    _lib.check(_lib.load().pulse_traj_step(ctypes.byref(a), _stream()), "pulse_traj_step")
    return out
