"""CPU, build container only: the step composition of oracle/motion_oracle.py:OracleMotionEnv (the CPU twin the GPU env is checked
against) versus the REFERENCE's own HumanoidIm method bodies -- _compute_reward, _compute_reset (incl. the cycle_motion restart),
_compute_task_obs (incl. fut_tracks) and the motion-library cache -- run on a stub task over the reference's MotionLibBase query
methods.  Same pre-step state in, same rewards / flags / observations / clock mutations out, bit for bit."""
import types

import pytest
import torch

from oracle import refload
from oracle.motion_oracle import OracleMotionEnv, OracleMotionLib
from pulse_amd import synthetic as syn

pytestmark = pytest.mark.skipif(not refload.available(), reason="reference checkout not mounted")


def _ref_motion_lib(tabs):
    lib = refload.motion_lib_class()()
    for k in ("gts", "grs", "lrs", "gvs", "gavs", "dvs", "length_starts"):
        setattr(lib, k, tabs[k])
    lib._motion_lengths, lib._motion_fps, lib._motion_dt = tabs["motion_lengths"], tabs["motion_fps"], tabs["motion_dt"]
    lib._motion_num_frames, lib.num_bodies, lib._device = tabs["motion_num_frames"], syn.NUM_BODIES, "cpu"
    total, m = tabs["gts"].shape[0], tabs["motion_lengths"].shape[0]
    lib._motion_aa, lib._motion_bodies, lib._motion_limb_weights = torch.zeros(total, 72), torch.zeros(m, 17), torch.zeros(m, 10)
    return lib


@pytest.mark.parametrize("cycle_motion,time_steps,zof", [(False, 1, None), (False, 3, None), (True, 1, None),
                                                         # zero_out_far (phc_kp_pnn_iccv.yaml:24,36-37: obs_v 7; phc_shape_*_iccv.yaml: obs_v 6)
                                                         (False, 1, (6, False)), (False, 1, (7, False)), (True, 1, (6, True)), (False, 1, (7, True)),
                                                         (True, 1, "xp"),       # cycle_motion_xp (humanoid_im.py:1133-1134)
                                                         # occl_training (:778-784, 827-831, 1178-1183) with a fresh random mask per step, alone and
                                                         # on top of zero_out_far
                                                         (False, 1, "occl6"), (False, 1, "occl7"), (False, 1, "occl6zof")])
def test_step_composition_matches_reference_methods(cycle_motion, time_steps, zof):
    xp = zof == "xp"
    occl = isinstance(zof, str) and zof.startswith("occl")
    if occl:
        zof = {"occl6": None, "occl7": (7, False), "occl6zof": (6, False)}[zof]
        occl_v7_plain = zof == (7, False)
    if xp:
        zof = None
    obs_v, zof_train = zof if zof else (6, False)
    f = refload.humanoid_im_methods()
    n, dt, frames = 48, 2.0 / 60.0, 12
    g = syn.make_generator(31)
    tabs = syn.synthetic_motion_library(g, n, 10, 24)                      # short clips: motions run out within a few steps
    noise = torch.randn(frames, n, 24, 13, generator=g) * torch.tensor([0.03] * 3 + [0.04] * 4 + [0.15] * 3 + [0.3] * 3)
    noise[:, ::7, 13, 0:3] += 1.0                                          # some envs drift away -> early termination
    if zof:                                                                # whole humanoids off the reference: 0.3 .. 7 m, some moving closer
        shift = torch.linspace(0.3, 7.0, n // 3)[None, :, None, None] * (1.0 - 0.05 * torch.arange(frames)[:, None, None, None])
        noise[:, ::3, :, 0:2] += shift
    bank = {"rb": noise, "dof_force": 50.0 * torch.randn(frames, n, 69, generator=g), "dof_pos": 0.02 * torch.randn(frames, n, 69, generator=g),
            "dof_vel": 0.1 * torch.randn(frames, n, 69, generator=g)}
    track = list(range(24))
    twin = OracleMotionEnv(OracleMotionLib(tabs), bank, torch.arange(n), torch.zeros(n, 3), syn.RESET_BODY_IDS, track, dt,
                           time_steps=time_steps, cycle_motion=cycle_motion, max_episode_length=300, obs_v=obs_v, zero_out_far=bool(zof),
                           zero_out_far_train=zof_train, close_distance=0.3, far_distance=2.5, zero_out_far_steps=4, cycle_motion_xp=xp)
    lib = _ref_motion_lib(tabs)
    starts = OracleMotionLib(tabs).sample_time_interval(torch.arange(n), generator=g)
    twin.reset(torch.arange(n), starts, far_uniforms=torch.rand(n, 2, generator=g))
    seen_far = seen_inside = seen_clipped = 0
    checked_cycle = 0
    for step in range(10):
        pre = {k: getattr(twin, k).clone() for k in ("progress", "start", "start_off", "offset", "cycle_counter", "point_goal")}
        cyc = OracleMotionLib(tabs).sample_time_interval(torch.arange(n), generator=g)
        far_u = torch.rand(n, 2, generator=g)
        if occl:
            twin.occl_idx = torch.rand(n, 24, generator=g) < 0.3
        obs_t, rew_t, reset_t, info_t = twin.step(cycle_start_times=cyc, far_uniforms=far_u, xp_uniforms=far_u)
        rb, fidx = twin.rb, twin.frame
        task = types.SimpleNamespace(
            _rigid_body_pos=rb[..., 0:3], _rigid_body_rot=rb[..., 3:7], _rigid_body_vel=rb[..., 7:10], _rigid_body_ang_vel=rb[..., 10:13],
            _humanoid_root_states=rb[:, 0], num_envs=n, device="cpu", humanoid_shapes=torch.zeros(n, 17), _fut_tracks=time_steps > 1,
            _num_traj_samples=time_steps, _traj_sample_timestep=1.0 / 30, progress_buf=pre["progress"] + 1, dt=dt,
            _motion_start_times=pre["start"].clone(), _motion_start_times_offset=pre["start_off"].clone(), _sampled_motion_ids=torch.arange(n),
            _global_offset=pre["offset"].clone(), _motion_lib=lib, ref_motion_cache={}, _track_bodies_id=torch.tensor(track), obs_v=obs_v,
            _has_upright_start=True, zero_out_far=bool(zof), zero_out_far_train=zof_train, close_distance=0.3, far_distance=2.5,
            _point_goal=pre["point_goal"].clone(), _occl_training=occl, random_occlu_idx=twin.occl_idx, _fut_tracks_dropout=False,
            ref_body_pos=torch.zeros(n, 24, 3), ref_body_vel=torch.zeros(n, 24, 3), ref_body_rot=torch.zeros(n, 24, 4),
            ref_body_pos_subset=torch.zeros(n, 24, 3), ref_dof_pos=torch.zeros(n, 69), dof_force_tensor=bank["dof_force"][fidx], _dof_vel=twin.dof_vel,
            reward_specs={"k_pos": 100, "k_rot": 10, "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1},
            _full_body_reward=True, power_reward=True, power_coefficient=0.0005, rew_buf=torch.zeros(n), reward_raw=torch.zeros(n, 4),
            max_episode_length=300, cycle_motion=cycle_motion, cycle_motion_xp=xp,
            _cycle_counter=torch.clamp_min(pre["cycle_counter"] - 1, 0) if (cycle_motion or zof_train) else pre["cycle_counter"].clone(),
            _sample_time=lambda ids: cyc[ids], reset_buf=torch.zeros(n, dtype=torch.int64), _terminate_buf=torch.zeros(n, dtype=torch.int64),
            _contact_forces=torch.zeros(n, 24, 3), _contact_body_ids=torch.tensor([7, 3]), _reset_bodies_id=torch.tensor(syn.RESET_BODY_IDS),
            _enable_early_termination=True, _termination_distances=torch.full((24,), 0.25), strict_eval=False)
        for k, fn in f.items():
            setattr(task, k, types.MethodType(fn, task))
        if zof_train or xp:                # the far restart of a cycled motion draws torch.rand twice (humanoid_im.py:1139-1140), cycle_motion_xp once
            draws = iter([far_u] if xp else [far_u[:, 0], far_u[:, 1]])            # ((k, 2), :1134): replay far_u there
            ended = (pre["progress"] + 1) * dt + pre["start"] + pre["start_off"] >= tabs["motion_lengths"][torch.arange(n)]
            f["_compute_reset"].__globals__["torch"] = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
            f["_compute_reset"].__globals__["torch"].rand = lambda *a, **kw: next(draws)[ended]
        try:
            task._compute_reward(None)
            task._compute_reset()
        finally:
            if zof_train or xp:
                f["_compute_reset"].__globals__["torch"] = torch
        task_obs = task._compute_task_obs()
        if zof:
            assert torch.equal(task._point_goal, twin.point_goal), f"point goal step {step}"
            seen_far += int((twin.point_goal > 0.3).sum())
            seen_inside += int((twin.point_goal <= 0.3).sum())
            seen_clipped += int((twin.point_goal > 2.5).sum())
        assert torch.equal(task.rew_buf, rew_t), f"reward step {step}"
        assert torch.equal(task.reward_raw, info_t["reward_raw"])
        assert torch.equal(task.reset_buf, reset_t) and torch.equal(task._terminate_buf, info_t["terminate"]), f"flags step {step}"
        assert torch.equal(task_obs, obs_t[:, 358:]), f"task obs step {step}"
        assert torch.equal(task._motion_start_times, twin.start) and torch.equal(task._motion_start_times_offset, twin.start_off)
        assert torch.equal(task._global_offset, twin.offset) and torch.equal(task._cycle_counter, twin.cycle_counter)
        checked_cycle += int((twin.cycle_counter == 60).sum())
        ids = torch.nonzero(reset_t).flatten()
        twin.reset(ids, OracleMotionLib(tabs).sample_time_interval(torch.arange(n), generator=g), far_uniforms=torch.rand(n, 2, generator=g))
    if zof:
        assert seen_far > 0 and seen_inside > 0 and seen_clipped > 0, "the far / inside / direction-only branches were not all taken"
    if zof_train or xp:
        assert (twin.offset != 0).any()
    if cycle_motion:
        assert checked_cycle > 0, "no motion was cycled in place"
    elif not zof_train:
        assert (twin.offset == 0).all()


def test_amp_window_matches_reference_methods():
    """OracleAmpHistory vs HumanoidAMP's own _compute_amp_observations / _update_hist_amp_obs / _init_amp_obs(_ref|_default) bodies."""
    from oracle.motion_oracle import OracleAmpHistory
    f = refload.humanoid_amp_methods()
    n, S, dt = 21, 10, 2.0 / 60.0
    g = syn.make_generator(8)
    tabs = syn.synthetic_motion_library(g, n, 10, 24)
    lib = _ref_motion_lib(tabs)
    key = torch.tensor([7, 3, 22, 17])
    # the shipped SMPL configuration: has_dof_subset True -> the 19 joints that are not toes / hands (humanoid.py:396-421)
    joints19 = [j for j in range(23) if j not in (3, 7, 17, 22)]
    subset = torch.tensor([3 * j + k for j in joints19 for k in range(3)])
    twin = OracleAmpHistory(OracleMotionLib(tabs), torch.arange(n), S, dt, key, dof_subset=subset)
    W = 196
    amp_buf = torch.zeros(n, S, W)
    task = types.SimpleNamespace(
        humanoid_type="smpl", dof_subset=subset, _amp_obs_buf=amp_buf, _curr_amp_obs_buf=amp_buf[:, 0], _hist_amp_obs_buf=amp_buf[:, 1:],
        _num_amp_obs_steps=S, dt=dt, device="cpu", _key_body_ids=key, _local_root_obs=True, _amp_root_height_obs=True, _has_dof_subset=True,
        _has_shape_obs_disc=False, _has_limb_weight_obs_disc=False, _has_upright_start=True, amp_obs_v=1, humanoid_shapes=torch.zeros(n, 17),
        humanoid_limb_and_weights=torch.zeros(n, 10), _motion_lib=lib, ref_motion_cache={}, gym=None, sim=None)
    for k, fn in f.items():
        setattr(task, k, types.MethodType(fn, task))

    class _OverlapChecked:
        """The history view with the partial-overlap check of the PyTorch the reference targets: `hist[:] = buf[:, 0:S-1]` raises
        there ("some elements of the input tensor and the written-to tensor refer to a single memory location") and
        _update_hist_amp_obs falls back to its `.clone()` form (humanoid_amp.py:624-627).  torch 2.10 on the CPU silently performs
        the overlapping copy instead (a smeared, undefined result), which is not what the reference computes on its own stack."""

        def __init__(self, view, whole):
            self.view, self.whole = view, whole

        def __setitem__(self, idx, val):
            lo, hi = self.whole.data_ptr(), self.whole.data_ptr() + self.whole.numel() * 4
            if isinstance(val, torch.Tensor) and lo <= val.data_ptr() < hi:
                raise RuntimeError("unsupported operation: some elements of the input tensor and the written-to tensor refer to a single memory location")
            self.view[idx] = val

        def __getitem__(self, idx):
            return self.view[idx]

    task._hist_amp_obs_buf = _OverlapChecked(amp_buf[:, 1:], amp_buf)

    def load_state(rb, dp, dv):
        task._rigid_body_pos, task._rigid_body_rot, task._rigid_body_vel, task._rigid_body_ang_vel = rb[..., 0:3], rb[..., 3:7], rb[..., 7:10], rb[..., 10:13]
        task._dof_pos, task._dof_vel = dp.clone(), dv.clone()                  # the reference zeroes toe / hand dofs IN PLACE

    starts = OracleMotionLib(tabs).sample_time_interval(torch.arange(n), generator=g)
    for step in range(6):
        rb = syn.rigid_body_state(g, n)
        dp, dv = 0.5 * torch.randn(n, 69, generator=g), torch.randn(n, 69, generator=g)
        load_state(rb, dp, dv)
        if step == 0:
            env_ids = torch.arange(n)
        elif step == 3:
            env_ids = torch.tensor([1, 4, 5, 17])
            starts = OracleMotionLib(tabs).sample_time_interval(torch.arange(n), generator=g)
        else:
            env_ids = None
        if env_ids is not None:                                                # reset: _init_amp_obs with reference-state init
            task._reset_default_env_ids, task._reset_ref_env_ids = [], env_ids
            task._reset_ref_motion_ids, task._reset_ref_motion_times = torch.arange(n)[env_ids], starts[env_ids]
            task._init_amp_obs(env_ids)
            twin.reset(env_ids, rb, dp, dv, starts, from_motion=True)
        else:                                                                  # HumanoidAMP.post_physics_step (:194-210)
            task._update_hist_amp_obs()
            task._compute_amp_observations()
            twin.step(rb, dp, dv)
        assert torch.equal(amp_buf, twin.buf), f"step {step}"
    # default init (no motion to look back into)
    task._reset_default_env_ids, task._reset_ref_env_ids = torch.tensor([0, 2]), []
    task._init_amp_obs(torch.tensor([0, 2]))
    twin.reset(torch.tensor([0, 2]), rb, dp, dv, starts, from_motion=False)
    assert torch.equal(amp_buf, twin.buf)


def test_small_step_options_match_reference_bodies():
    """fut_tracks_dropout inside _compute_task_obs (humanoid_im.py:804-810) and res_action's _action_to_pd_targets (:1096-1101): the
    oracle's restatements against the reference's method bodies, the torch.rand draw replayed."""
    from oracle import env_oracle as E
    f = refload.humanoid_im_methods()
    n, T = 29, 3
    g = syn.make_generator(12)
    tabs = syn.synthetic_motion_library(g, n, 10, 24)
    lib = _ref_motion_lib(tabs)
    rb = syn.rigid_body_state(g, n)
    track = list(range(24))
    u = torch.rand(n, T, generator=g)
    u[0, 1] = 0.01                                                         # at least one dropped and one kept sample
    u[1] = 0.9
    task = types.SimpleNamespace(
        _rigid_body_pos=rb[..., 0:3], _rigid_body_rot=rb[..., 3:7], _rigid_body_vel=rb[..., 7:10], _rigid_body_ang_vel=rb[..., 10:13], num_envs=n,
        device="cpu", humanoid_shapes=torch.zeros(n, 17), _fut_tracks=True, _num_traj_samples=T, _traj_sample_timestep=1.0 / 30,
        progress_buf=torch.randint(0, 5, (n,), generator=g), dt=1.0 / 30, _motion_start_times=torch.rand(n, generator=g) * 0.2,
        _motion_start_times_offset=torch.zeros(n), _sampled_motion_ids=torch.arange(n), _global_offset=torch.zeros(n, 3), _motion_lib=lib,
        ref_motion_cache={}, _track_bodies_id=torch.tensor(track), obs_v=6, _has_upright_start=True, zero_out_far=False, _occl_training=False,
        _fut_tracks_dropout=False, ref_body_pos=torch.zeros(n, 24, 3), ref_body_vel=torch.zeros(n, 24, 3), ref_body_rot=torch.zeros(n, 24, 4),
        ref_body_pos_subset=torch.zeros(n, 24, 3), ref_dof_pos=torch.zeros(n, 69))
    for k, fn in f.items():
        setattr(task, k, types.MethodType(fn, task))
    clean = task._compute_task_obs()
    task._fut_tracks_dropout = True
    fake = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    fake.rand = lambda *a, **kw: u
    f["_compute_task_obs"].__globals__["torch"] = fake
    try:
        dropped = task._compute_task_obs()
    finally:
        f["_compute_task_obs"].__globals__["torch"] = torch
    assert torch.equal(dropped, E.fut_tracks_dropout(clean, u, T))
    assert (dropped.view(n, T, -1)[0, 1] == 0).all() and torch.equal(dropped[1], clean[1])
    # res_action
    act = torch.randn(n, 69, generator=g) * 2.0
    t2 = types.SimpleNamespace(_res_action=True, ref_dof_pos=torch.randn(n, 69, generator=g), _pd_action_scale=torch.rand(69, generator=g) + 0.5,
                               _pd_action_offset=torch.zeros(69), _dof_pos=torch.randn(n, 69, generator=g))
    want = f["_action_to_pd_targets"](t2, act)
    got = E.res_action_pd_targets(t2.ref_dof_pos, t2._pd_action_scale, act, t2._dof_pos)
    assert torch.equal(want, got)
    assert (got == t2._dof_pos + torch.pi / 2).any() or (got == t2._dof_pos - torch.pi / 2).any()      # the clamp is exercised


@pytest.mark.parametrize("cls", ["HumanoidSpeed", "HumanoidStrike"])
def test_power_usage_reward_matches_reference_bodies(cls):
    """[r6] power_usage_reward: the block lives in the METHOD bodies of HumanoidSpeed / HumanoidStrike._compute_reward (humanoid_speed.py:225-238,
    humanoid_strike.py:186-198).  The bodies are executed on a stub over several steps (the accumulator carries over) and compared with
    oracle.task_oracle.power_usage_reward on top of the pinned jit rewards: reward, reward_raw layout and the accumulator, bit for bit."""
    import types
    from oracle import task_oracle as TO
    from pulse_amd import synthetic as syn
    body = refload.task_reward_methods()[cls]
    g = torch.Generator().manual_seed(5)
    n = 37
    dof_names = syn.SMPL_BODY_NAMES[1:]
    left, right = TO.side_dof_indexes(dof_names, lower_only=False)
    left_lo, right_lo = TO.side_dof_indexes(dof_names, lower_only=True)
    assert len(left) == len(right) == 9 and len(left_lo) == len(right_lo) == 4          # L_/R_ Hip Knee Ankle Toe Thorax Shoulder Elbow Wrist Hand
    st = types.SimpleNamespace(num_envs=n, dt=1 / 30, power_reward=cls == "HumanoidSpeed", power_usage_reward=True, power_coefficient=0.0005,
                               power_usage_coefficient=0.0025, left_indexes=left, right_indexes=right, left_lower_indexes=left_lo,
                               right_lower_indexes=right_lo, power_acc=torch.zeros(n, 2), rew_buf=torch.zeros(n), _near_dist=1.4,
                               _strike_body_ids=torch.tensor([23]))
    acc = torch.zeros(n, 2)
    for step in range(6):
        root = torch.randn(n, 13, generator=g)
        root[:, 3:7] = torch.nn.functional.normalize(root[:, 3:7], dim=-1)
        st._humanoid_root_states, st._prev_root_pos = root, root[:, 0:3] - 0.03 * torch.randn(n, 3, generator=g)
        st._tar_speed = torch.rand(n, generator=g) * 3
        st._target_states = torch.randn(n, 13, generator=g)
        st._target_states[:, 3:7] = torch.nn.functional.normalize(st._target_states[:, 3:7], dim=-1)
        st._rigid_body_vel = torch.randn(n, 24, 3, generator=g)
        st.dof_force_tensor, st._dof_vel = torch.randn(n, 69, generator=g) * 30, torch.randn(n, 69, generator=g)
        st.progress_buf = torch.randint(0, 9, (n,), generator=g)
        body(st, None)
        if cls == "HumanoidSpeed":
            base = TO.speed_reward(root[:, 0:3], st._prev_root_pos, root[:, 3:7], st._tar_speed, st.dt)
            pw = -0.0005 * (st.dof_force_tensor * st._dof_vel).abs().sum(-1)
            pw[st.progress_buf <= 3] = 0
            pur = TO.power_usage_reward(st.dof_force_tensor, st._dof_vel, acc, st.progress_buf, left, right, 0.0025)
            assert torch.equal(st.rew_buf, base + pw + pur)
            assert torch.equal(st.reward_raw, torch.stack([base, pw, pur], dim=-1))
        else:
            base = TO.strike_reward(st._target_states[:, 0:3], st._target_states[:, 3:7], root, st._prev_root_pos, st._rigid_body_vel[:, 23], st.dt, 1.4)
            pur = TO.power_usage_reward(st.dof_force_tensor, st._dof_vel, acc, st.progress_buf, left_lo, right_lo, 0.0025)
            assert torch.equal(st.rew_buf, base + pur)
        assert torch.equal(st.power_acc, acc) and (pur != 0).any() and (pur[st.progress_buf <= 3] == 0).all()
