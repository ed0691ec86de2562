"""CPU: the oracle's restatement of the ``zero_out_far`` branch (oracle/env_oracle.py: zero_out_far_refs, point_goal_reward,
im_reward_zero_out_far, post_physics_zero_out_far) reproduces, bit for bit, tests/golden/env_zero_out_far.npz -- written by the
reference's own HumanoidIm._compute_reward / _compute_reset / _compute_task_obs METHOD BODIES and compute_point_goal_reward
(oracle/gen_golden.py: gen_env_zero_out_far; phc/env/tasks/humanoid_im.py:763-777, 814-826, 870-887, 1158-1176, 1577-1582)."""
import os

import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from pulse_amd import synthetic as syn

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_zero_out_far.npz"))
t = lambda k: torch.from_numpy(Z[k])

CASES = [(6, list(range(24)), "v6"), (7, syn.VR_TRACK_BODY_IDS, "v7_vr"), (7, list(range(24)), "v7"), (8, list(range(24)), "v8"),
         (9, list(range(24)), "v9"), (6, syn.VR_TRACK_BODY_IDS, "v6_vr")]


def same(a, key):
    assert a.shape == Z[key].shape, (key, a.shape, Z[key].shape)
    assert np.array_equal(a.numpy(), Z[key], equal_nan=True), key


def refs(which):
    return {k: t(f"ref_{which}_{k}") for k in ("pos", "rot", "vel", "ang")}


def test_point_goal_reward():
    rb = t("rb")
    r, raw = E.point_goal_reward(t("point_goal_prev"), torch.norm(rb[:, 0, 0:3] - t("ref_now_pos")[:, 0], dim=-1))
    same(r, "point_goal_reward")
    assert (Z["point_goal_reward"] == 3.0).any(), "the 1/3 m clamp is not exercised"


@pytest.mark.parametrize("obs_v,ids,tag", CASES)
@pytest.mark.parametrize("close,far,dtag", [(0.25, 3.0, ""), (0.5, 1.5, "_c05_f15")])
def test_post_physics_zero_out_far(obs_v, ids, tag, close, far, dtag):
    o = E.post_physics_zero_out_far(t("rb"), refs("now"), refs("next"), t("point_goal_prev"), t("dof_force"), t("dof_vel"), t("progress"),
                                    t("pass_time"), syn.RESET_BODY_IDS, ids, torch.full((1, 24), 0.25), obs_v=obs_v, close_distance=close,
                                    far_distance=far)
    same(o["obs"][:, 358:], f"task_obs_{tag}{dtag}")
    same(o["point_goal"], f"point_goal_{tag}{dtag}")
    same(o["rew"], "reward")
    same(o["raw"], "reward_raw")
    same(o["reset"], "reset")
    same(o["terminate"], "terminate")


def test_golden_covers_every_branch():
    pg = Z["point_goal_v6"]
    assert (pg <= 0.25).sum() >= 10 and ((pg > 0.25) & (pg <= 3.0)).sum() >= 10 and (pg > 3.0).sum() >= 10
    raw = Z["reward_raw"]
    inside = np.linalg.norm(Z["rb"][:, 0, 0:3] - Z["ref_now_pos"][:, 0], axis=-1) <= 0.25
    assert (raw[~inside, 1:4] == 0).all() and (raw[inside, 1:4] > 0).any()      # outside: point-goal term only
    # a far env's difference blocks are exactly zero for bodies 1.. (its own state is the reference)
    far = pg > 0.25
    dpos = Z["task_obs_v6"][:, 0:72].reshape(-1, 24, 3)
    assert (dpos[far, 1:] == 0).all() and (dpos[~far, 1:] != 0).any()
