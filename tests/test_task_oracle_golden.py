"""CPU: the downstream-task oracle (oracle/task_oracle.py: speed / reach / strike observations, rewards, resets -- SURVEY.md 8(f) rank 4,
oracle-first for the next round's kernels) reproduces the golden produced by the reference's own TorchScript functions bit for bit."""
import os

import numpy as np
import torch

from oracle import task_oracle as T

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tasks.npz"))
t = lambda k: torch.from_numpy(Z[k])


def same(a, key):
    assert np.array_equal(a.numpy(), Z[key], equal_nan=True), key


def test_speed_reach_strike_observations_and_rewards():
    root, dt = t("root_states"), float(Z["dt"])
    same(T.speed_observations(root, t("tar_speed")), "speed_obs")
    same(T.speed_reward(root[:, 0:3], t("prev_root_pos"), root[:, 3:7], t("tar_speed"), dt), "speed_rew")
    same(T.location_observations(root, t("tar_pos")), "loc_obs")
    same(T.reach_reward(t("reach_body_pos"), root[:, 3:7], t("tar_pos"), 1.0, dt), "reach_rew")
    ts = t("tar_states")
    same(T.strike_observations(root, ts), "strike_obs")
    same(T.strike_reward(ts[:, 0:3], ts[:, 3:7], root, t("prev_root_pos"), t("strike_body_vel"), dt, 1.4), "strike_rew")
    assert (Z["strike_rew"] == 1).any() and (Z["strike_rew"] < 1).any()            # both the success and the shaping branch


def test_humanoid_resets():
    n = Z["progress"].shape[0]
    reset0 = torch.zeros(n, dtype=torch.long)
    args = (reset0, t("progress"), t("contact"), t("contact_ids"), t("body_pos"))
    r, term = T.humanoid_reset(*args, 300.0, True, t("term_h"))
    same(r, "reset"); same(term, "terminated")
    r2, _ = T.humanoid_reset(*args, 300.0, False, t("term_h"))
    same(r2, "reset_noearly")
    r3, t3 = T.strike_reset(*args, t("tar_contact"), t("strike_ids"), 300.0, True, t("term_h"))
    same(r3, "strike_reset"); same(t3, "strike_terminated")
    assert Z["terminated"].sum() > 0 and (Z["reset"] != Z["terminated"]).any() and Z["strike_terminated"].sum() >= Z["terminated"].sum()
    assert not Z["terminated"][:2].any()                                            # progress <= 1 never terminates
