"""GPU: HumanoidIm with the HBM-resident motion library as its reference source, in lockstep with
the CPU twin oracle/motion_oracle.py:OracleMotionEnv: observations, rewards, reset / terminate flags and the AMP demo
windows over several dozen steps with partial resets."""
import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from oracle.motion_oracle import OracleMotionEnv, OracleMotionLib
from pulse_amd import configs
from pulse_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _twin(env, n, seed, time_steps=1):
    task = env.task
    tabs = syn.synthetic_motion_library(syn.make_generator(seed + 5, 0), min(n, 1024))
    lib = OracleMotionLib(tabs)
    bank = {k: v.cpu() for k, v in task.sim.bank.items()}
    return OracleMotionEnv(lib, bank, task._sampled_motion_ids.cpu(), task._global_offset.cpu(), task._reset_bodies_id.cpu().long(),
                           task._track_bodies_id.cpu().long(), task.dt, time_steps=time_steps, traj_dt=task._traj_sample_timestep,
                           obs_v=task.obs_v, cycle_motion=task.cycle_motion, max_episode_length=task.max_episode_length)


@pytest.mark.parametrize("n,overrides", [(67, {}), (40, {"fut_tracks": True, "numTrajSamples": 3}), (33, {"obs_v": 7, "trackBodies": ["Head", "L_Hand", "R_Hand"]}),
                                         (52, {"cycle_motion": True, "episode_length": 45})])
def test_motion_lib_env_lockstep_with_cpu_twin(dev, n, overrides):
    seed, horizon = 321, 24
    env, _ = configs.make_env(n, horizon, dev, seed=seed, reference="motion_lib", env_overrides=overrides)
    task = env.task
    twin = _twin(env, n, seed, time_steps=task._num_traj_samples)
    obs = env.reset()
    obs = obs["obs"] if isinstance(obs, dict) else obs
    o_ref = twin.reset(torch.arange(n), task._motion_start_times.cpu())
    assert task._motion_start_times.max() > 0                                     # random state init
    np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=3e-5, rtol=1e-5)
    g = torch.Generator().manual_seed(1)
    n_done = 0
    for step in range(60):
        act = torch.randn(n, 69, generator=g).to(dev)
        obs, rew, done, info = env.step(act)
        obs = obs["obs"] if isinstance(obs, dict) else obs
        o_ref, r_ref, d_ref, i_ref = twin.step(cycle_start_times=task._last_cycle_start.cpu() if task.cycle_motion else None)
        np.testing.assert_allclose(rew.cpu().numpy(), r_ref.numpy(), atol=2e-5, rtol=1e-5, err_msg=f"reward step {step}")
        assert torch.equal(done.cpu(), d_ref), f"reset flags step {step}"
        assert torch.equal(info["terminate"].cpu(), i_ref["terminate"]), f"terminate step {step}"
        np.testing.assert_allclose(info["reward_raw"].cpu().numpy(), i_ref["reward_raw"].numpy(), atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5, err_msg=f"obs step {step}")
        ids = torch.nonzero(d_ref).flatten()
        n_done += ids.numel()
        obs = env.reset(ids.to(dev))
        obs = obs["obs"] if isinstance(obs, dict) else obs
        o_ref = twin.reset(ids, task._motion_start_times.cpu())
        np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5, err_msg=f"obs after reset {step}")
        assert torch.equal(task.progress_buf.cpu(), twin.progress)
        assert torch.equal(task._cycle_counter.cpu(), twin.cycle_counter)
        np.testing.assert_allclose(task._global_offset.cpu().numpy(), twin.offset.numpy(), atol=1e-5)
    assert n_done > 0, "the lockstep run never exercised a reset"
    if task.cycle_motion:
        assert (task._motion_start_times_offset != 0).any(), "no motion was cycled in place"


def test_amp_demo_windows_from_motion_lib(dev):
    n, seed = 32, 77
    env, _ = configs.make_env(n, 8, dev, seed=seed, env_kind="amp", reference="motion_lib")
    task = env.task
    env.reset()
    k = 48
    state = task._clock_gen.get_state()
    demo = env.fetch_amp_obs_demo(k)
    assert demo.shape == (k, task.get_num_amp_obs())
    # replay the draws, evaluate the windows on the CPU
    task._clock_gen.set_state(state)
    lib = task._motion_lib
    ids = lib.sample_motions(k, generator=task._clock_gen).cpu()
    t0 = lib.sample_time_interval(ids.to(dev), generator=task._clock_gen).cpu()       # HumanoidIm._sample_time (humanoid_im.py:652-654)
    assert torch.allclose(t0 * 30, torch.round(t0 * 30), atol=1e-4)                     # demo windows start on the 1/30 s grid
    tabs = syn.synthetic_motion_library(syn.make_generator(seed + 5, 0), min(n, 1024))
    orc = OracleMotionLib(tabs)
    s = task._num_amp_obs_steps
    times = (t0.unsqueeze(-1) + (-task.dt * torch.arange(0, s))).view(-1)
    st = orc.get_motion_state(ids.repeat_interleave(s), times)
    key = task._key_body_ids.cpu().long()
    subset = torch.tensor([3 * int(j) + kk for j in task._amp_joint_ids.cpu() for kk in range(3)])
    want = E.amp_obs_smpl(st["root_pos"], st["root_rot"], st["root_vel"], st["root_ang_vel"], st["dof_pos"], st["dof_vel"], st["rg_pos"][:, key],
                          dof_subset=subset)
    np.testing.assert_allclose(demo.cpu().numpy(), want.view(k, -1).numpy(), atol=3e-5, rtol=1e-5)


def test_agent_epoch_on_motion_lib_env(dev):
    torch.manual_seed(3)
    ag, _ = configs.make_agent("cfg1", device=dev, seed=5, reference="motion_lib")
    for e in range(2):
        ag.epoch_num = e + 1
        info = ag.train_epoch()
    assert all(torch.isfinite(torch.as_tensor(v, dtype=torch.float32)).all() for v in info["actor_loss"])
    r = ag.experience_buffer.phys["rewards"]
    assert torch.isfinite(r).all()
    raw = ag.vec_env.env.task.reward_raw
    assert (raw[:, :4] > 0.02).all()                          # the tracked humanoid earns imitation reward on every term
    assert ag.experience_buffer.phys["dones"].sum() > 0


def test_in_kernel_reference_equals_array_reference(dev):
    """pulse_im_step blending the reference from the packed library == the same launch fed by pulse_motion_state's arrays
    (bit for bit: same device functions), and the in-kernel clock == progress += 1 / pass_time done outside."""
    from pulse_amd import ops
    from pulse_amd._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS
    n = 131
    env, _ = configs.make_env(n, 8, dev, seed=9, reference="motion_lib", env_overrides={"fut_tracks": True, "numTrajSamples": 2})
    task = env.task
    env.reset()
    for _ in range(3):
        env.step(torch.zeros(n, 69, device=dev))
        env.reset(torch.nonzero(task.reset_buf).flatten())
    task.sim.simulate_and_refresh()
    lib = task._motion_lib
    what = PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS
    prog0 = task.progress_buf.clone()
    # (a) arrays path: advance the clock by hand, query the library, feed the arrays
    prog = prog0 + 1
    t = prog * task.dt + task._motion_start_times + task._motion_start_times_offset
    pass_time = t >= task._motion_len_env
    kw = dict(offset=task._global_offset, progress=prog, dt=task.dt, start_times=task._motion_start_times, start_offsets=task._motion_start_times_offset)
    now = lib.query(task._sampled_motion_ids, step_shift=0, **kw)
    nxt = lib.query(task._sampled_motion_ids, step_shift=1, time_steps=2, traj_dt=task._traj_sample_timestep, **kw)
    ref = lambda r: {"pos": r["rg_pos"], "rot": r["rb_rot"], "vel": r["body_vel"], "ang": r["body_ang_vel"]}
    common = dict(time_steps=2, dof_force=task.sim.dof_force, dof_vel=task.sim.dof_vel, cycle_counter=task._cycle_counter,
                  track_ids=task._track_bodies_id, reset_ids=task._reset_bodies_id, term_dist=task._termination_distances,
                  specs=task.reward_specs, power_coef=task.power_coefficient, power_reward=task.power_reward)
    a = ops.im_step(task.sim.rigid_body_state, what=what, ref_now=ref(now), ref_next=ref(nxt), progress=prog, pass_time=pass_time, **common)
    # (b) fused path
    task.progress_buf.copy_(prog0)
    clock, motion = task._motion_kwargs(inc=1)
    b = ops.im_step(task.sim.rigid_body_state, what=what, progress=task.progress_buf, clock=clock, motion=motion, **common)
    for k in ("obs", "rew", "rew_raw", "reset", "terminate"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(task.progress_buf, prog) and torch.equal(task._pass_time, pass_time)
    one = lib.query(task._sampled_motion_ids, step_shift=1, with_records=True, **kw)
    assert torch.equal(task._track["rb_records"], one["rb_records"]) and torch.equal(task._track["dof_pos"], one["dof_pos"])
    assert torch.equal(task._track["dof_vel"], one["dof_vel"])


def test_reset_mode_matches_separate_ops(dev):
    n = 77
    env, _ = configs.make_env(n, 8, dev, seed=10, reference="motion_lib")
    task = env.task
    env.reset()
    lib, sim = task._motion_lib, task.sim
    for _ in range(2):
        env.step(torch.zeros(n, 69, device=dev))
    mask = torch.rand(n, device=dev) < 0.4
    before = {k: v.clone() for k, v in dict(rb=sim.rigid_body_state, dp=sim.dof_pos, dv=sim.dof_vel, st=task._motion_start_times,
                                            pg=task.progress_buf).items()}
    task.reset_buf.fill_(1)
    task._terminate_buf.fill_(1)
    state = task._clock_gen.get_state()
    task.reset_masked(mask)
    task._clock_gen.set_state(state)
    phase = torch.zeros(n, device=dev).uniform_(0.0, 1.0, generator=task._clock_gen)
    st = torch.where(mask, ((phase * task._motion_len_env) / (1 / 30)).long() * (1 / 30), before["st"])      # sample_time_interval
    assert torch.equal(task._motion_start_times, st)
    assert torch.equal(task.progress_buf, before["pg"] * (~mask))
    assert torch.equal(task.reset_buf, (~mask).long()) and torch.equal(task._terminate_buf, (~mask).long())
    assert (task._global_offset[mask] == 0).all() and (task._motion_start_times_offset[mask] == 0).all() and (task._cycle_counter[mask] == 0).all()
    want = lib.query(task._sampled_motion_ids, st, task._global_offset, with_records=True)
    assert torch.equal(sim.rigid_body_state, torch.where(mask[:, None, None], want["rb_records"], before["rb"]))
    assert torch.equal(sim.dof_pos, torch.where(mask[:, None], want["dof_pos"], before["dp"]))
    assert torch.equal(sim.dof_vel, torch.where(mask[:, None], want["dof_vel"], before["dv"]))


def test_amp_window_lockstep_with_cpu_twin(dev):
    """The (N, 10, 196) AMP observation window of the env (slot 0 = simulated frame, shifted every step, re-initialised from the
    motion before the start time on reference-state resets) vs oracle/motion_oracle.py:OracleAmpHistory, which is pinned bit for
    bit to HumanoidAMP's own methods (tests/test_oracle_env_vs_reference_methods.py)."""
    from oracle.motion_oracle import OracleAmpHistory
    n, seed = 45, 19
    env, _ = configs.make_env(n, 12, dev, seed=seed, env_kind="amp", reference="motion_lib")
    task = env.task
    tabs = syn.synthetic_motion_library(syn.make_generator(seed + 5, 0), min(n, 1024))
    assert task._num_amp_obs_per_step == 196 and task.get_num_amp_obs() == 1960            # robot/smpl_humanoid.yaml: has_dof_subset
    subset = torch.tensor([3 * int(j) + k for j in task._amp_joint_ids.cpu() for k in range(3)])
    twin = OracleAmpHistory(OracleMotionLib(tabs), task._sampled_motion_ids.cpu(), task._num_amp_obs_steps, task.dt, task._key_body_ids.cpu().long(),
                            dof_subset=subset)
    sim = task.sim
    state = lambda: (sim.rigid_body_state.cpu().clone(), sim.dof_pos.cpu().clone(), sim.dof_vel.cpu().clone())
    env.reset()
    twin.reset(torch.arange(n), *state(), task._motion_start_times.cpu(), from_motion=True)
    np.testing.assert_allclose(task._amp_obs_buf.cpu().numpy(), twin.buf.numpy(), atol=3e-5, rtol=1e-5)
    assert not torch.equal(task._amp_obs_buf[:, 1], task._amp_obs_buf[:, 0])            # history comes from the motion, not copies of frame 0
    n_reset = 0
    for step in range(30):
        obs, rew, done, info = env.step(torch.zeros(n, 69, device=dev))
        want = twin.step(*state())
        np.testing.assert_allclose(info["amp_obs"].cpu().numpy(), want.numpy(), atol=3e-5, rtol=1e-5, err_msg=f"amp_obs step {step}")
        ids = torch.nonzero(done).flatten()
        n_reset += ids.numel()
        env.reset(ids)
        twin.reset(ids.cpu(), *state(), task._motion_start_times.cpu(), from_motion=True)
        np.testing.assert_allclose(task._amp_obs_buf.cpu().numpy(), twin.buf.numpy(), atol=3e-5, rtol=1e-5, err_msg=f"window after reset {step}")
    assert n_reset > 0


def test_step_returns_a_fresh_observation_like_the_reference(dev):
    """vec_task.py:152-157 returns torch.clamp(obs_buf, ...): a caller may keep the tensor across steps.  The wrapper does the same
    unless an agent that copies it right away opts into the aliased buffer (alias_obs)."""
    n = 16
    env, _ = configs.make_env(n, 8, dev, seed=5, reference="motion_lib")
    env.reset()
    o1, *_ = env.step(torch.zeros(n, 69, device=dev))
    keep = o1.clone()
    o2, *_ = env.step(torch.zeros(n, 69, device=dev))
    assert o1.data_ptr() != env.task.obs_buf.data_ptr() and o2.data_ptr() != o1.data_ptr()
    assert torch.equal(o1, keep) and not torch.equal(o1, o2)
    env.alias_obs = True
    o3, *_ = env.step(torch.zeros(n, 69, device=dev))
    assert o3.data_ptr() == env.task.obs_buf.data_ptr()


def test_fused_amp_window_kernels_equal_the_op_by_op_path(dev, monkeypatch):
    """pulse_amp_obs in history mode (shift + current frame + window copy in one launch) and pulse_amp_hist_init (history of the reset
    envs from the motion, one launch) against the op-by-op path they replace (clone / copy_ shift, motion query + AMP frames of all
    N x 9 rows + where): bit-identical windows over 25 steps with resets, and the window lands in the caller's sink rows."""
    n, seed = 52, 23

    def run(fused):
        monkeypatch.setenv("PULSE_AMP_FUSED", "1" if fused else "0")
        env, _ = configs.make_env(n, 12, dev, seed=seed, env_kind="amp", reference="motion_lib")
        task = env.task
        assert task._amp_fused == fused
        env.reset()
        out = [task._amp_obs_buf.clone()]
        sink = torch.full((n, 3, 1964), float("nan"), device=dev)                 # rows of a (N, T, pitch) experience buffer, slot 1
        for step in range(25):
            if fused and step % 2 == 0:
                task.set_amp_obs_sink(sink[:, 1])
            obs, rew, done, info = env.step(torch.zeros(n, 69, device=dev))
            if fused and step % 2 == 0:
                assert info["amp_obs"].data_ptr() == sink[:, 1].data_ptr()          # the extras ARE the sink rows
                assert torch.equal(sink[:, 1, :1960], task._amp_obs_buf.view(n, -1)) and torch.isnan(sink[:, 0]).all() and torch.isnan(sink[:, 1, 1960:]).all()
            out.append(info["amp_obs"].clone())
            env.reset(torch.nonzero(done).flatten())
            out.append(task._amp_obs_buf.clone())
        return out
    a, b = run(True), run(False)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x.reshape(n, -1), y.reshape(n, -1)), f"window {i} differs"
