"""GPU: the remaining observation variants of the fused env step (obs_v 1 / 2 / 3 / 8 / 9, self_obs_v 2 / 3, non-upright start,
remove_base_rot -- SURVEY.md 8 rows a7 / f4) and the downstream-task kernel (speed / reach / strike observations, rewards and
compute_humanoid_reset) against the goldens written by the reference's own functions (tests/golden/env_variants.npz, tasks.npz)."""
import os

import numpy as np
import pytest
import torch

from pulse_amd import ops, synthetic as syn
from pulse_amd._lib import TASK_OBS, TASK_RESET, TASK_REWARD

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
Z = np.load(os.path.join(GOLD, "env_variants.npz"))
ZT = np.load(os.path.join(GOLD, "tasks.npz"))


def close(a, key, z=Z, atol=1e-5):
    a = a.detach().cpu().numpy()
    assert a.shape == z[key].shape, (key, a.shape, z[key].shape)
    np.testing.assert_allclose(a, z[key], atol=atol, rtol=1e-5, err_msg=key)


@pytest.fixture(scope="module")
def inp(dev):
    t = lambda k: torch.from_numpy(Z[k]).to(dev)
    rb = t("rb")
    n, T = rb.shape[0], int(Z["T"])
    return {"rb": rb, "n": n, "T": T, "ref": [t("ref_pos"), t("ref_rot"), t("ref_vel"), t("ref_ang")], "t": t}


def test_remove_base_rot(inp):
    close(ops.remove_base_rot(inp["rb"][:, 0, 3:7].contiguous()), "remove_base_rot", atol=2e-7)


@pytest.mark.parametrize("upright", [True, False])
def test_imitation_observation_variants(inp, upright):
    rb, n, T, ref = inp["rb"], inp["n"], inp["T"], inp["ref"]
    bp, br, bv, ba = rb[..., 0:3], rb[..., 3:7], rb[..., 7:10], rb[..., 10:13]
    tag = "" if upright else "_noup"
    full, vr = list(range(24)), syn.VR_TRACK_BODY_IDS
    fns = {1: ops.compute_imitation_observations, 3: ops.compute_imitation_observations_v3, 6: ops.compute_imitation_observations_v6}
    for ids, itag in ((full, ""), (vr, "_vr")):
        cur = [x[:, ids].contiguous() for x in (bp, br, bv, ba)]
        rf = [x[:, ids].contiguous() for x in ref]
        for ver, fn in fns.items():
            close(fn(bp[:, 0], br[:, 0], *cur, *rf, T, upright), f"v{ver}_T{T}{itag}{tag}")
        close(ops.compute_imitation_observations_v9(bp[:, 0], br[:, 0], *cur, rf[0], rf[1], rf[2][:, 0].contiguous(), rf[3][:, 0].contiguous(),
                                                    T, upright), f"v9_T{T}{itag}{tag}")
    one = lambda x: x.view(n, T, *x.shape[1:])[:, 0].contiguous()
    r1 = [one(x) for x in ref]
    cur = [x.contiguous() for x in (bp, br, bv, ba)]
    close(ops.compute_imitation_observations_v8(bp[:, 0], br[:, 0], *cur, *r1, 1, upright), f"v8_T1{tag}")
    close(ops.compute_imitation_observations_v7(bp[:, 0], br[:, 0], bp[:, vr].contiguous(), bv[:, vr].contiguous(), r1[0][:, vr].contiguous(),
                                                r1[2][:, vr].contiguous(), 1, upright), f"v7_T1_vr{tag}")
    dsel = lambda d: d.reshape(-1, 23, 3)[:, [i - 1 for i in full[1:]], :].contiguous()
    close(ops.compute_imitation_observations_v2(bp[:, 0], br[:, 0], *cur, dsel(inp["t"]("dof_pos")), *r1, dsel(inp["t"]("ref_dof_pos")), 1, upright),
          f"v2_T1{tag}")


@pytest.mark.parametrize("upright", [True, False])
def test_self_observation_variants(inp, upright):
    rb, t = inp["rb"], inp["t"]
    bp, br, bv, ba = (rb[..., 0:3].contiguous(), rb[..., 3:7].contiguous(), rb[..., 7:10].contiguous(), rb[..., 10:13].contiguous())
    tag = "" if upright else "_noup"
    for lro, ltag in ((True, ""), (False, "_globalroot")):
        close(ops.compute_humanoid_observations_smpl_max(bp, br, bv, ba, None, None, lro, True, upright), f"self_obs{ltag}{tag}")
        close(ops.compute_humanoid_observations_smpl_max_v3(bp, br, bv, ba, t("force_sensor"), None, None, lro, True, upright), f"self_obs_v3{ltag}{tag}")
    h = t("rb_hist")
    close(ops.compute_humanoid_observations_smpl_max_v2(h[..., 0:3], h[..., 3:7], h[..., 7:10], h[..., 10:13], None, None, True, True, upright,
                                                        False, False, int(Z["T"])), f"self_obs_v2{tag}")


def test_early_termination_flag(dev):
    n = 70
    g = syn.make_generator(5)
    d = syn.env_step_inputs(g, n)
    rb = d["rb"].to(dev)
    ref = {k: v.to(dev) for k, v in d["ref_now"].items()}
    kw = dict(what=ops.PULSE_IM_RESET, ref_now=ref, progress=d["progress"].to(dev), pass_time=d["pass_time"].to(dev),
              reset_ids=syn.RESET_BODY_IDS, term_dist=torch.full((24,), 0.25, device=dev))
    on = ops.im_step(rb, enable_early_termination=True, **kw)
    off = ops.im_step(rb, enable_early_termination=False, **kw)
    assert on["terminate"].sum() > 0 and off["terminate"].sum() == 0
    assert torch.equal(off["reset"], d["pass_time"].to(dev).long())          # only time-outs reset (humanoid_im.py:1610-1613)


def _task_inputs(dev):
    t = lambda k: torch.from_numpy(ZT[k]).to(dev)
    n = ZT["progress"].shape[0]
    rb = torch.zeros(n, 24, 13, device=dev)
    rb[:, :, 0:3] = t("body_pos")
    rb[:, 0] = t("root_states")
    rb[:, 23, 0:3] = t("reach_body_pos")
    rb[:, 23, 7:10] = t("strike_body_vel")
    return t, n, rb


def test_task_observations_and_rewards(dev):
    t, n, rb = _task_inputs(dev)
    dt = float(ZT["dt"])
    o = ops.task_step("speed", rb, what=TASK_OBS | TASK_REWARD, prev_root_pos=t("prev_root_pos"), dt=dt, tar_speed=t("tar_speed"))
    close(o["obs"], "speed_obs", ZT)
    close(o["rew"], "speed_rew", ZT)
    o = ops.task_step("reach", rb, what=TASK_OBS | TASK_REWARD, dt=dt, tar_pos=t("tar_pos"), reach_body_id=23)
    close(o["obs"], "loc_obs", ZT)
    close(o["rew"], "reach_rew", ZT)
    o = ops.task_step("strike", rb, what=TASK_OBS | TASK_REWARD, prev_root_pos=t("prev_root_pos"), dt=dt, tar_states=t("tar_states"))
    close(o["obs"], "strike_obs", ZT)
    close(o["rew"], "strike_rew", ZT)
    # task observation written at its column offset of a pitched observation row; other columns untouched
    row = torch.full((n, 384), 7.0, device=dev)
    ops.task_step("speed", rb, what=TASK_OBS, tar_speed=t("tar_speed"), obs=row, obs_offset=358)
    close(row[:, 358:361], "speed_obs", ZT)
    assert (row[:, :358] == 7.0).all() and (row[:, 361:] == 7.0).all()


def test_task_resets_bit_exact(dev):
    t, n, rb = _task_inputs(dev)
    rb[:, :, 0:3] = t("body_pos")
    common = dict(what=TASK_RESET, contact_forces=t("contact"), contact_body_ids=t("contact_ids"), termination_heights=t("term_h"),
                  progress=t("progress"), max_episode_length=300.0)
    o = ops.task_step("speed", rb, enable_early_termination=True, **common)
    assert np.array_equal(o["reset"].cpu().numpy(), ZT["reset"]) and np.array_equal(o["terminate"].cpu().numpy(), ZT["terminated"])
    o = ops.task_step("reach", rb, enable_early_termination=False, **common)
    assert np.array_equal(o["reset"].cpu().numpy(), ZT["reset_noearly"]) and o["terminate"].sum() == 0
    o = ops.task_step("strike", rb, enable_early_termination=True, tar_contact_forces=t("tar_contact"), strike_body_ids=t("strike_ids"), **common)
    assert np.array_equal(o["reset"].cpu().numpy(), ZT["strike_reset"]) and np.array_equal(o["terminate"].cpu().numpy(), ZT["strike_terminated"])
    # power term of the speed reward (humanoid_speed.py:211-218)
    g = torch.Generator().manual_seed(3)
    df, dv = torch.randn(n, 69, generator=g).to(dev) * 50, torch.randn(n, 69, generator=g).to(dev)
    o = ops.task_step("speed", rb, what=TASK_REWARD, prev_root_pos=t("prev_root_pos"), dt=float(ZT["dt"]), tar_speed=t("tar_speed"),
                      dof_force=df, dof_vel=dv, progress=t("progress"), power_reward=True)
    pw = -0.0005 * (df * dv).abs().sum(-1)
    pw[t("progress") <= 3] = 0
    np.testing.assert_allclose(o["rew"].cpu().numpy(), ZT["speed_rew"] + pw.cpu().numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(o["rew_raw"][:, 1].cpu().numpy(), pw.cpu().numpy(), atol=1e-5, rtol=1e-5)


def test_shape_and_limb_weight_observation_rows(dev):
    """has_smpl_params / has_limb_weight_params (humanoid.py:1724-1728, 1843-1847) vs the golden written by the reference's functions;
    also through the env (robot has_shape_obs / has_weight_obs -> + 11 / + 10 columns, humanoid.py:657-661)."""
    S = np.load(os.path.join(GOLD, "env_shape_obs.npz"))
    g = lambda k: torch.from_numpy(S[k]).to(dev)
    rb = g("rb")
    bp, br, bv, ba = (rb[..., 0:3], rb[..., 3:7], rb[..., 7:10], rb[..., 10:13])
    sh, lw, fs = g("smpl_params"), g("limb_weights"), g("force_sensor")
    for up, tag in ((True, ""), (False, "_noup")):
        for name, hs, hl in (("both", True, True), ("shape", True, False), ("limb", False, True)):
            got = ops.compute_humanoid_observations_smpl_max(bp, br, bv, ba, sh, lw, True, True, up, hs, hl)
            want = g(f"self_obs_{name}{tag}")
            assert got.shape == want.shape and (got - want).abs().max().item() < 2e-6, (name, tag)
            assert torch.equal(got[:, 358:], want[:, 358:])                          # the appended rows are copies
        got = ops.compute_humanoid_observations_smpl_max_v3(bp, br, bv, ba, fs, sh, lw, True, True, up, True, True)
        want = g(f"self_obs_v3_both{tag}")
        assert got.shape == want.shape and (got - want).abs().max().item() < 2e-6 and torch.equal(got[:, 358:], want[:, 358:])
    with pytest.raises(Exception):                                                    # the reference's _v2 raises for these options
        ops.im_step(torch.zeros(4, 2, 24, 13, device=dev), what=1, self_obs_version=2, smpl_params=torch.zeros(4, 11, device=dev))
    from pulse_amd import configs
    agent, _ = configs.make_agent("cfg1", device=dev, seed=3, env_overrides={"has_shape_obs": True, "has_weight_obs": True})
    task = agent.vec_env.env.task
    assert task.get_self_obs_size() == 379 and task.num_obs == 379 + 576
    agent.init_tensors()
    agent.env_reset()
    task.step(torch.zeros(task.num_envs, task.num_actions, device=dev))
    assert torch.equal(task.obs_buf[:, 358:369], task.humanoid_shapes[:, :11]) and torch.equal(task.obs_buf[:, 369:379], task.humanoid_limb_and_weights)
    assert torch.isfinite(task.obs_buf).all()
