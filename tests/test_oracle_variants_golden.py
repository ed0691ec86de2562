"""CPU: the oracle's restatement of the remaining observation variants (obs_v 1 / 2 / 3 / 8 / 9, self_obs_v 2 / 3, remove_base_rot and
the non-upright forms -- SURVEY.md 8 rows a7 and f4) reproduces, bit for bit, the golden written by the reference's own
TorchScript functions (oracle/gen_golden.py: gen_env_variants; phc/env/tasks/humanoid_im.py:1222-1540, humanoid.py:1616-1849)."""
import os

import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from pulse_amd import synthetic as syn

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_variants.npz"))
t = lambda k: torch.from_numpy(Z[k])


def same(a, key):
    assert a.shape == Z[key].shape, (key, a.shape, Z[key].shape)
    assert np.array_equal(a.numpy(), Z[key], equal_nan=True), key


def _inputs():
    rb = t("rb")
    n, T = rb.shape[0], int(Z["T"])
    return rb, n, T, (t("ref_pos"), t("ref_rot"), t("ref_vel"), t("ref_ang"))


def test_remove_base_rot():
    same(E.remove_base_rot(t("rb")[:, 0, 3:7].contiguous()), "remove_base_rot")


@pytest.mark.parametrize("upright", [True, False])
def test_imitation_observation_variants(upright):
    rb, n, T, ref = _inputs()
    bp, br, bv, ba = E.split_rb(rb)
    tag = "" if upright else "_noup"
    full, vr = list(range(24)), syn.VR_TRACK_BODY_IDS
    for ids, itag in ((full, ""), (vr, "_vr")):
        cur = [x[:, ids].contiguous() for x in (bp, br, bv, ba)]
        rf = [x[:, ids].contiguous() for x in ref]
        for ver in (1, 3, 6, 9):
            same(E.im_obs_variant(ver, bp[:, 0], br[:, 0], *cur, *rf, time_steps=T, upright=upright), f"v{ver}_T{T}{itag}{tag}")
    one = lambda x: x.view(n, T, *x.shape[1:])[:, 0].contiguous()
    r1 = [one(x) for x in ref]
    same(E.im_obs_variant(8, bp[:, 0], br[:, 0], bp, br, bv, ba, *r1, time_steps=1, upright=upright), f"v8_T1{tag}")
    cur = [x[:, vr].contiguous() for x in (bp, br, bv, ba)]
    same(E.im_obs_variant(7, bp[:, 0], br[:, 0], *cur, *[x[:, vr].contiguous() for x in r1], time_steps=1, upright=upright), f"v7_T1_vr{tag}")
    dsel = lambda d: d.reshape(-1, 23, 3)[:, [i - 1 for i in full[1:]], :].contiguous()
    same(E.im_obs_variant(2, bp[:, 0], br[:, 0], bp, br, bv, ba, *r1, time_steps=1, upright=upright, dof_pos=dsel(t("dof_pos")),
                          ref_dof_pos=dsel(t("ref_dof_pos"))), f"v2_T1{tag}")


@pytest.mark.parametrize("upright", [True, False])
def test_self_observation_variants(upright):
    rb, n, T, _ = _inputs()
    bp, br, bv, ba = E.split_rb(rb)
    tag = "" if upright else "_noup"
    for lro, ltag in ((True, ""), (False, "_globalroot")):
        same(E.self_obs_smpl_max_general(bp, br, bv, ba, lro, True, upright), f"self_obs{ltag}{tag}")
        same(E.self_obs_smpl_max_general(bp, br, bv, ba, lro, True, upright, force_sensor=t("force_sensor")), f"self_obs_v3{ltag}{tag}")
    h = t("rb_hist")
    same(E.self_obs_smpl_max_v2(h[..., 0:3].contiguous(), h[..., 3:7].contiguous(), h[..., 7:10].contiguous(), h[..., 10:13].contiguous(),
                                True, True, upright), f"self_obs_v2{tag}")
    with pytest.raises(NotImplementedError):
        E.self_obs_smpl_max_v2(h[..., 0:3], h[..., 3:7], h[..., 7:10], h[..., 10:13], False, True, upright)
