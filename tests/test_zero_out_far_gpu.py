"""GPU: the ``zero_out_far`` branch of the fused env step (pulse_im_step: masked reference in the task observation, point-goal reward,
_point_goal bookkeeping) -- SURVEY.md section 8 rows a9 / a10 callers, reference phc/env/tasks/humanoid_im.py:763-777, 814-826, 870-887,
932-946, 1135-1142, 1577-1582.

  * against tests/golden/env_zero_out_far.npz, written by the reference's own _compute_reward / _compute_reset / _compute_task_obs
    method bodies (oracle/gen_golden.py: gen_env_zero_out_far);
  * HumanoidIm on the motion library in lockstep with the CPU twin (oracle/motion_oracle.py, itself pinned to those method bodies by
    tests/test_oracle_env_vs_reference_methods.py), incl. zero_out_far_train's far starts and far restarts of cycled motions;
  * a phc_kp_pnn_iccv-shaped env (obs_v 7, zero_out_far, cycle_motion) at 4096 envs against the oracle.
Tolerances: floats 1e-5 (north_star), flags and indices exact."""
import os

import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from oracle.motion_oracle import OracleMotionEnv, OracleMotionLib
from pulse_amd import configs, ops
from pulse_amd import synthetic as syn
from pulse_amd._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS

pytestmark = pytest.mark.gpu

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_zero_out_far.npz"))
ALL = PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS
CASES = [(6, list(range(24)), "v6"), (7, syn.VR_TRACK_BODY_IDS, "v7_vr"), (7, list(range(24)), "v7"), (8, list(range(24)), "v8"),
         (9, list(range(24)), "v9"), (6, syn.VR_TRACK_BODY_IDS, "v6_vr")]


def _golden_launch(dev, obs_v, ids, close, far, what=ALL, point_goal=None):
    t = lambda k: torch.from_numpy(Z[k]).to(dev)
    ref = lambda w: {k: t(f"ref_{w}_{k}") for k in ("pos", "rot", "vel", "ang")}
    pg = t("point_goal_prev").clone() if point_goal is None else point_goal
    out = ops.im_step(t("rb"), what=what, ref_now=ref("now"), ref_next=ref("next"), dof_force=t("dof_force"), dof_vel=t("dof_vel"),
                      progress=t("progress"), pass_time=t("pass_time"), track_ids=ids, reset_ids=syn.RESET_BODY_IDS,
                      term_dist=torch.full((24,), 0.25, device=dev), obs_version=obs_v,
                      zero_out_far={"point_goal": pg, "close_distance": close, "far_distance": far})
    return out, pg


@pytest.mark.parametrize("obs_v,ids,tag", CASES)
@pytest.mark.parametrize("close,far,dtag", [(0.25, 3.0, ""), (0.5, 1.5, "_c05_f15")])
def test_fused_step_vs_reference_method_golden(dev, obs_v, ids, tag, close, far, dtag):
    out, pg = _golden_launch(dev, obs_v, ids, close, far)
    task = out["obs"][:, 358:].cpu().numpy()
    err = np.abs(task - Z[f"task_obs_{tag}{dtag}"]).max()
    assert err <= 1e-5, f"task obs {tag}{dtag}: {err}"
    # a far env's own state is its reference: the difference blocks of bodies 1.. are EXACT zeros, as in the reference
    far_env = Z[f"point_goal_{tag}{dtag}"] > close
    jt = len(ids)
    dpos = task[:, :3 * jt].reshape(-1, jt, 3)
    assert (dpos[far_env, 1:] == 0).all()
    np.testing.assert_allclose(pg.cpu().numpy(), Z[f"point_goal_{tag}{dtag}"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["rew"].cpu().numpy(), Z["reward"], rtol=0, atol=1e-5)          # reward used the OLD point goal
    np.testing.assert_allclose(out["rew_raw"].cpu().numpy(), Z["reward_raw"], rtol=0, atol=1e-5)
    assert np.array_equal(out["reset"].cpu().numpy(), Z["reset"]) and np.array_equal(out["terminate"].cpu().numpy(), Z["terminate"])


def test_stage_order_is_the_references(dev):
    """Reward alone must not touch _point_goal; the observation stage alone must write it; one fused launch == the two in sequence."""
    rew_only, pg = _golden_launch(dev, 6, list(range(24)), 0.25, 3.0, what=PULSE_IM_REWARD)
    assert np.array_equal(pg.cpu().numpy(), Z["point_goal_prev"])
    obs_only, pg2 = _golden_launch(dev, 6, list(range(24)), 0.25, 3.0, what=PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS, point_goal=pg)
    fused, pg3 = _golden_launch(dev, 6, list(range(24)), 0.25, 3.0)
    assert torch.equal(pg2, pg3) and torch.equal(fused["rew"], rew_only["rew"]) and torch.equal(fused["obs"], obs_only["obs"])
    assert torch.equal(fused["rew_raw"], rew_only["rew_raw"])


def test_rejects_what_the_reference_cannot_do(dev):
    from pulse_amd._lib import PulseLibraryError as PulseError
    t = lambda k: torch.from_numpy(Z[k]).to(dev)
    kw = dict(what=PULSE_IM_TASK_OBS, track_ids=list(range(24)), ref_next={k: t(f"ref_next_{k}") for k in ("pos", "rot", "vel", "ang")})
    with pytest.raises((PulseError, ValueError)):                                 # obs_v 1: no zero_out_far block in the reference
        ops.im_step(t("rb"), obs_version=1, zero_out_far={"point_goal": torch.zeros(83, device=dev)}, **kw)
    with pytest.raises(NotImplementedError):
        configs.make_env(8, 4, dev, reference="motion_lib", env_overrides={"zero_out_far": True, "fut_tracks": True})
    with pytest.raises(NotImplementedError):                                      # far starts move the reference by a per-env offset
        configs.make_env(8, 4, dev, env_overrides={"zero_out_far": True, "zero_out_far_train": True})


def _twin(task, n, seed):
    tabs = syn.synthetic_motion_library(syn.make_generator(seed + 5, 0), min(n, 1024))
    bank = {k: v.cpu() for k, v in task.sim.bank.items()}
    return OracleMotionEnv(OracleMotionLib(tabs), bank, task._sampled_motion_ids.cpu(), task._global_offset.cpu(), task._reset_bodies_id.cpu().long(),
                           task._track_bodies_id.cpu().long(), task.dt, obs_v=task.obs_v, cycle_motion=task.cycle_motion,
                           max_episode_length=task.max_episode_length, zero_out_far=True, zero_out_far_train=task.zero_out_far_train,
                           close_distance=task.close_distance, far_distance=task.far_distance, zero_out_far_steps=task._zero_out_far_steps)


def _lockstep(dev, n, overrides, steps, seed=321, atol=2e-5):
    env, _ = configs.make_env(n, 24, dev, seed=seed, reference="motion_lib", env_overrides=dict({"zero_out_far": True}, **overrides))
    task = env.task
    # whole humanoids off their reference by 0.3 .. 7 m, drifting closer: inside / far / direction-only envs in every step
    frames = task.sim.bank["rb"].shape[0]
    k = len(range(0, n, 3))
    shift = torch.linspace(0.3, 7.0, k, device=dev)[None, :, None, None] * (1.0 - 0.03 * torch.arange(frames, device=dev)[:, None, None, None])
    task.sim.bank["rb"][:, ::3, :, 0:2] += shift
    twin = _twin(task, n, seed)
    far_u = lambda: task._last_far_uniforms.cpu() if task.zero_out_far_train else None
    obs = env.reset()
    o_ref = twin.reset(torch.arange(n), task._motion_start_times.cpu(), far_uniforms=far_u())
    np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(task._point_goal.cpu().numpy(), twin.point_goal.numpy(), atol=1e-5)
    seen = {"far": 0, "inside": 0, "dir": 0, "done": 0}
    for step in range(steps):
        obs, rew, done, info = env.step(torch.zeros(n, 69, device=dev))
        o_ref, r_ref, d_ref, i_ref = twin.step(cycle_start_times=task._last_cycle_start.cpu() if task.cycle_motion else None,
                                               far_uniforms=far_u() if task.cycle_motion else None)
        np.testing.assert_allclose(rew.cpu().numpy(), r_ref.numpy(), atol=atol, rtol=1e-5, err_msg=f"reward step {step}")
        np.testing.assert_allclose(info["reward_raw"].cpu().numpy(), i_ref["reward_raw"].numpy(), atol=atol, rtol=1e-5)
        assert torch.equal(done.cpu(), d_ref) and torch.equal(info["terminate"].cpu(), i_ref["terminate"]), f"flags step {step}"
        np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5, err_msg=f"obs step {step}")
        np.testing.assert_allclose(task._point_goal.cpu().numpy(), twin.point_goal.numpy(), atol=1e-5)
        pg = twin.point_goal
        seen["far"] += int((pg > task.close_distance).sum()); seen["inside"] += int((pg <= task.close_distance).sum())
        seen["dir"] += int((pg > task.far_distance).sum())
        ids = torch.nonzero(d_ref).flatten()
        seen["done"] += ids.numel()
        obs = env.reset(ids.to(dev))
        if ids.numel():
            o_ref = twin.reset(ids, task._motion_start_times.cpu(), far_uniforms=far_u())
            np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5, err_msg=f"obs after reset {step}")
        assert torch.equal(task._cycle_counter.cpu(), twin.cycle_counter)
        np.testing.assert_allclose(task._global_offset.cpu().numpy(), twin.offset.numpy(), atol=1e-5)
    # (with the VR subset the distance runs from the root to the HEAD's reference, humanoid_im.py:816: "inside" needs a generous close_distance)
    assert seen["far"] and seen["dir"] and seen["done"] and (seen["inside"] or int(task._track_bodies_id[0]) != 0), seen
    return task, twin


@pytest.mark.parametrize("n,overrides", [(67, {"zero_out_far_train": False}),
                                         (45, {"zero_out_far_train": False, "obs_v": 7, "trackBodies": ["Head", "L_Hand", "R_Hand"], "close_distance": 1.0,
                                               "far_distance": 3.0}),
                                         (52, {"zero_out_far_train": True, "zero_out_far_steps": 5}),
                                         (52, {"zero_out_far_train": True, "zero_out_far_steps": 5, "cycle_motion": True, "episode_length": 45})])
def test_env_lockstep_with_cpu_twin(dev, n, overrides):
    task, twin = _lockstep(dev, n, overrides, steps=40)
    if overrides.get("zero_out_far_train"):
        assert (task._global_offset[:, 0:2] != 0).any()
        r = task._global_offset[:, 0:2].norm(dim=-1)
        assert (r <= 5.0 + 1e-4).all() or task.cycle_motion                    # far starts lie within the 5 m disc (cycled restarts add the root offset)


def test_phc_kp_pnn_iccv_shaped_env_at_4096(dev):
    """The env section of phc/data/cfg/env/phc_kp_pnn_iccv.yaml as the reference would read it (obs_v 7, zero_out_far True,
    zero_out_far_train False, cycle_motion True (the later of its two entries), power_reward True) at BASELINE's 4096 envs."""
    env_cfg = {"obs_v": 7, "zero_out_far": True, "zero_out_far_train": False, "cycle_motion": True, "power_reward": True, "fut_tracks": False,
               "numTrajSamples": 3, "trajSampleTimestepInv": 3, "enableTaskObs": True, "stateInit": "Random", "hybridInitProb": 0.5,
               "numAMPObsSteps": 10, "controlFrequencyInv": 2, "terminationHeight": 0.15, "enableEarlyTermination": True, "terminationDistance": 0.25,
               "key_bodies": ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"], "reset_bodies": syn.RESET_BODY_NAMES, "has_upright_start": True,
               "has_dof_subset": True, "shape_resampling_interval": 500, "getup_udpate_epoch": 78750, "hard_negative": False, "kp_scale": 1,
               "episode_length": 300}
    task, twin = _lockstep(dev, 4096, env_cfg, steps=6, atol=1e-5)
    assert task.obs_v == 7 and task.num_obs == 358 + 9 * 24
