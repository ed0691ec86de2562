"""CPU: the oracle restatement reproduces the REAL reference outputs (tests/golden) bit for bit."""
import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from oracle import rotations as R
from pulse_amd import synthetic as syn


def same(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    return np.array_equal(a, b, equal_nan=True)


def test_rotations_bit_exact(golden):
    g = golden("rotations.npz")
    q, p, v, e, t = (g.t(k) for k in "qpvet")
    assert same(R.qmul(q, p), g.np("quat_mul"))
    assert same(R.qconj(q), g.np("quat_conjugate"))
    assert same(R.qrot(q, v), g.np("my_quat_rotate"))
    ang, ax = R.q_to_angle_axis(q)
    assert same(ang, g.np("quat_to_angle")) and same(ax, g.np("quat_to_axis"))
    assert same(R.q_to_exp_map(q), g.np("quat_to_exp_map"))
    assert same(R.q_to_tan_norm(q), g.np("quat_to_tan_norm"))
    assert same(R.exp_map_to_q(e), g.np("exp_map_to_quat"))
    assert same(R.slerp(q, p, t), g.np("slerp"))
    assert same(R.heading(q), g.np("calc_heading"))
    assert same(R.heading_q(q), g.np("calc_heading_quat"))
    assert same(R.heading_q_inv(q), g.np("calc_heading_quat_inv"))
    # edge rows really exercise the masked branches
    assert g.np("quat_to_angle")[0] == 0 and g.np("quat_to_angle")[4] == 0
    assert not np.isnan(g.np("slerp")).any()


def test_isaacgym_boundary_cross_check(golden):
    """The 3P isaacgym quat_mul (unpinned) agrees with poselib's in-tree 16-multiply statement."""
    g = golden("rotations.npz")
    np.testing.assert_allclose(R.qmul(g.t("q"), g.t("p")).numpy(), g.np("poselib_quat_mul"), atol=5e-7, rtol=0)


def _env(g):
    rb = g.t("rb")
    rn = {k: g.t("ref_now_" + k) for k in ("pos", "rot", "vel", "ang")}
    rx = {k: g.t("ref_next_" + k) for k in ("pos", "rot", "vel", "ang")}
    return rb, rn, rx


def test_env_functions_bit_exact(golden):
    g = golden("env_im.npz")
    rb, rn, rx = _env(g)
    bp, br, bv, ba = E.split_rb(rb)
    assert same(E.self_obs_smpl_max(bp, br, bv, ba), g.np("self_obs"))
    assert same(E.self_obs_smpl_max(bp, br, bv, ba, local_root_obs=False), g.np("self_obs_global_root"))
    assert same(E.im_obs_v6(bp[:, 0], br[:, 0], bp, br, bv, ba, rx["pos"], rx["rot"], rx["vel"], rx["ang"]), g.np("task_obs_v6"))
    tb = syn.VR_TRACK_BODY_IDS
    assert same(E.im_obs_v7(bp[:, 0], br[:, 0], bp[:, tb], bv[:, tb], rx["pos"][:, tb], rx["vel"][:, tb]), g.np("task_obs_v7_vr"))
    assert same(E.im_obs_v6(bp[:, 0], br[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb], rx["pos"][:, tb], rx["rot"][:, tb],
                            rx["vel"][:, tb], rx["ang"][:, tb]), g.np("task_obs_v6_vr"))
    rew, raw = E.im_reward(bp, br, bv, ba, rn["pos"], rn["rot"], rn["vel"], rn["ang"])
    assert same(rew, g.np("reward_im")) and same(raw, g.np("reward_raw_im"))
    rew, raw = E.im_reward_full(rb, rn["pos"], rn["rot"], rn["vel"], rn["ang"], g.t("dof_force"), g.t("dof_vel"), g.t("progress"))
    assert same(rew, g.np("reward")) and same(raw, g.np("reward_raw"))


def test_post_physics_matches_reference_pieces(golden):
    g = golden("env_im.npz")
    rb, rn, rx = _env(g)
    out = E.post_physics(rb, rn, rx, g.t("dof_force"), g.t("dof_vel"), g.t("progress"), g.t("pass_time"),
                         syn.RESET_BODY_IDS, list(range(24)), torch.full((1, 24), 0.25))
    assert same(out["obs"], np.concatenate([g.np("self_obs"), g.np("task_obs_v6")], -1))
    assert out["obs"].shape[1] == 934
    assert same(out["rew"], g.np("reward")) and same(out["raw"], g.np("reward_raw"))
    assert same(out["reset"], g.np("reset")) and same(out["terminate"], g.np("terminate"))
    assert g.np("terminate").sum() >= 1 and g.np("reset").sum() > g.np("terminate").sum()
    out = E.post_physics(rb, rn, rx, g.t("dof_force"), g.t("dof_vel"), g.t("progress"), g.t("pass_time"),
                         syn.RESET_BODY_IDS, list(range(24)), torch.full((1, 24), 0.25), use_mean=True)
    assert same(out["reset"], g.np("reset_mean")) and same(out["terminate"], g.np("terminate_mean"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_gae_bit_exact(golden, tag):
    g = golden("agent_math.npz")
    adv = E.gae(g.t(f"gae_{tag}_dones").float(), g.t(f"gae_{tag}_values"), g.t(f"gae_{tag}_rewards"),
                g.t(f"gae_{tag}_next_values"), 0.99, 0.95)
    assert same(adv, g.np(f"gae_{tag}_advs"))


def test_amp_observation_bit_exact(golden):
    g = golden("env_amp.npz")
    rb = g.t("rb")
    bp, br, bv, ba = E.split_rb(rb)
    key = list(g.np("key_body_ids"))
    args = (bp[:, 0], br[:, 0], bv[:, 0], ba[:, 0], g.t("dof_pos"), g.t("dof_vel"), bp[:, key])
    assert same(E.amp_obs_smpl(*args), g.np("amp_obs_full"))
    assert g.np("amp_obs_full").shape[1] == 232                               # 13 + 23*6 + 69 + 12 (humanoid_amp.py:299-303)
    sub = [3 * j + k for j in g.np("joints19") for k in range(3)]
    assert same(E.amp_obs_smpl(*args, dof_subset=sub, root_height_obs=False), g.np("amp_obs_subset19_noheight"))
    assert g.np("amp_obs_subset19_noheight").shape[1] == 195                  # env_pulse_amp.yaml:61 width
    assert same(E.amp_obs_smpl(*args, local_root_obs=False), g.np("amp_obs_global_root"))
    assert same(E.dof_to_obs_smpl(g.t("dof_pos")), g.np("dof_to_obs"))
