"""CPU: the oracle's restatement of the trajectory / terrain task (oracle/task_oracle.py) reproduces the vectors written by the reference's
own code (oracle/gen_golden.py: gen_terrain -- TrajGenerator, the TorchScript functions of humanoid_pedestrian_terrain.py / humanoid_traj.py
and the get_heights / get_center_heights / sample_height_points method bodies executed on stubs)."""
import math
import os

import numpy as np
import torch

from oracle import task_oracle as TO

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "terrain.npz"))
t = lambda k: torch.from_numpy(Z[k])


def test_trajectory_generator_and_sampling():
    n, V = t("verts").shape[0], t("verts").shape[1]
    dt = float(Z["traj_dt"])
    root = t("rb")[:, 0]
    # the deterministic half of TrajGenerator.reset from the recorded uniform draws
    dtheta = (2 * t("u_dtheta") - 1.0) * (2.0 * dt)
    sharp = math.pi * (2 * t("u_sharp") - 1.0)
    dtheta[t("u_sharp_mask")] = sharp[t("u_sharp_mask")]
    dtheta[:, 0] = math.pi * (2 * t("u_heading") - 1.0)
    dspeed = (2 * t("u_dspeed") - 1.0) * (2.0 * dt)
    dspeed[:, 0] = (3.0 - 0.0) * t("u_speed0") + 0.0
    verts = TO.traj_from_draws(root[:, 0:3], dtheta, dspeed, dt, 0.0, 3.0)
    assert torch.equal(verts, t("verts"))
    # and the drawing order itself: one seed, same trajectories
    torch.manual_seed(4321)
    assert torch.equal(TO.traj_generate(root[:, 0:3], V, dt, 2.0, 0.0, 3.0, 2.0, 0.02), t("verts"))
    prog = t("progress")
    assert torch.equal(TO.traj_calc_pos(t("verts"), torch.arange(n), prog * float(Z["dt"]), dt), t("tar_pos"))
    assert torch.equal(TO.fetch_traj_samples(t("verts"), prog, float(Z["dt"]), dt, 10, 0.5), t("traj_samples"))


def test_location_obs_reward_and_resets_bit_exact():
    root = t("rb")[:, 0]
    for up, tag in ((True, ""), (False, "_noup")):
        assert torch.equal(TO.traj_location_observations(root, t("traj_samples"), up), t(f"loc_obs{tag}"))
    assert torch.equal(TO.traj_location_observations(root, t("traj_samples"), True), t("traj_loc_obs"))
    assert torch.equal(TO.location_reward(root[:, 0:3], t("tar_pos")), t("loc_rew"))
    assert torch.equal(TO.location_reward(root[:, 0:3], t("tar_pos") + 0.03, fuzzy=True), t("loc_rew_fuzzy"))
    z0 = torch.zeros(root.shape[0], dtype=torch.long)
    r, term = TO.terrain_reset(z0, t("progress"), t("contact"), t("contact_ids"), t("body_pos"), t("far_tar_pos"), 300.0, 4.0, True)
    assert torch.equal(r, t("terrain_reset")) and torch.equal(term, t("terrain_terminated")) and 0 < int(term.sum()) < term.numel()
    r, _ = TO.terrain_reset(z0, t("progress"), t("contact"), t("contact_ids"), t("body_pos"), t("far_tar_pos"), 300.0, 4.0, False)
    assert torch.equal(r, t("terrain_reset_noearly"))
    r, term = TO.traj_reset(z0, t("progress"), t("contact"), t("contact_ids"), t("body_pos"), t("far_tar_pos"), 300.0, 4.0, True, t("term_h"))
    assert torch.equal(r, t("traj_reset")) and torch.equal(term, t("traj_terminated"))
    assert not torch.equal(t("traj_terminated"), t("terrain_terminated"))          # the two variants really differ on these inputs


def test_height_map_sampling_bit_exact():
    rb = t("rb")
    root, head = rb[:, 0], rb[:, 13]
    hp = torch.cat([t("height_points"), torch.zeros(1024, 1)], dim=1)
    cp = torch.cat([t("center_points"), torch.zeros(9, 1)], dim=1)
    hs = t("heightsamples")
    for up, tag in ((True, ""), (False, "_noup")):
        assert torch.equal(TO.terrain_heights(hs, head[:, 0:7], hp, 0.1, 0.005, up), t(f"heights{tag}"))
        assert torch.equal(TO.terrain_center_heights(hs, root, cp, 0.1, 0.005, up), t(f"center_heights{tag}"))
        obs = TO.terrain_task_obs(root, head[:, 0:7], t("traj_samples"), hs, hp, cp, upright=up)
        assert obs.shape == (rb.shape[0], 20 + 1024) and torch.equal(obs, t(f"task_obs{tag}"))
    assert len(np.unique(Z["heights"])) > 50                                        # a real relief, not a plane
