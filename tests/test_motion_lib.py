"""Reference-motion query (MotionLibBase.get_motion_state and friends).

CPU: the oracle restatement reproduces the golden produced by the reference's own methods bit for bit.
GPU: pulse_motion_state (through MotionLib) against the golden and, at larger sizes, against the oracle:
frame indices / blend / every lerp'd field bit-exact, slerp / exp-map fields to 2e-6 absolute."""
import os

import numpy as np
import pytest
import torch

from oracle.motion_oracle import OracleMotionLib
from pulse_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "motion_lib.npz")
EXACT = ("rg_pos", "body_vel", "body_ang_vel", "dof_vel", "root_pos", "root_vel", "root_ang_vel")
ROT = ("rb_rot", "root_rot", "dof_pos")


def _golden():
    z = np.load(GOLDEN)
    tabs = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("tab_")}
    return z, tabs


def test_oracle_reproduces_reference_golden():
    z, tabs = _golden()
    o = OracleMotionLib(tabs)
    ids, times, off = (torch.from_numpy(z[k]) for k in ("motion_ids", "motion_times", "offset"))
    r = o.get_motion_state(ids, times, off)
    for k in EXACT + ROT:
        assert np.array_equal(r[k].numpy(), z[k]), k
    base = tabs["length_starts"][ids].numpy()
    assert np.array_equal(r["frame_idx0"].numpy() - base, z["frame_idx0"])
    assert np.array_equal(r["frame_idx1"].numpy() - base, z["frame_idx1"])
    assert np.array_equal(r["blend"].numpy(), z["blend"])
    assert np.array_equal(o.get_motion_state(ids, times)["rg_pos"].numpy(), z["rg_pos_no_offset"])
    assert np.array_equal(o.get_root_pos_smpl(ids, times)["root_pos"].numpy(), z["root_pos_smpl"])
    assert np.array_equal(o.get_motion_num_steps().numpy(), z["num_steps"])
    # the golden covers the edge cases: t < 0, t past the end, exact frame times
    assert (z["motion_times"] < 0).any() and (z["blend"] == 0).any() and (z["frame_idx0"] == z["frame_idx1"]).any()


def test_synthetic_library_is_a_valid_clip_set():
    tabs = syn.synthetic_motion_library(syn.make_generator(3), 7, 10, 30)
    nf = tabs["motion_num_frames"]
    assert tabs["gts"].shape == (int(nf.sum()), 24, 3) and tabs["dvs"].shape[1:] == (23, 3)
    assert torch.allclose(tabs["grs"].norm(dim=-1), torch.ones(()), atol=1e-5)
    assert torch.equal(tabs["length_starts"], torch.cumsum(nf, 0) - nf)
    for v in tabs.values():
        assert torch.isfinite(v.float()).all()


def _compare(res, want, rot_atol=2e-6):
    for k in EXACT:
        assert torch.equal(res[k].cpu(), want[k]), k
    for k in ROT:
        np.testing.assert_allclose(res[k].cpu().numpy(), want[k].numpy(), atol=rot_atol, rtol=0, err_msg=k)


@pytest.mark.gpu
def test_motion_state_matches_reference_golden(dev):
    from pulse_amd.env.motion_lib import MotionLib
    z, tabs = _golden()
    lib = MotionLib.from_tables(tabs, dev)
    ids, times, off = (torch.from_numpy(z[k]).to(dev) for k in ("motion_ids", "motion_times", "offset"))
    res = lib.get_motion_state(ids, times, off)
    _compare(res, {k: torch.from_numpy(z[k]) for k in EXACT + ROT})
    fr = lib.query(ids, times, off, with_frames=True)
    assert np.array_equal(fr["frame_idx0"].cpu().numpy(), z["frame_idx0"])
    assert np.array_equal(fr["frame_idx1"].cpu().numpy(), z["frame_idx1"])
    assert np.array_equal(fr["blend"].cpu().numpy(), z["blend"])
    assert np.array_equal(lib.get_motion_state(ids, times)["rg_pos"].cpu().numpy(), z["rg_pos_no_offset"])
    assert np.array_equal(lib.get_root_pos_smpl(ids, times)["root_pos"].cpu().numpy(), z["root_pos_smpl"])
    assert np.array_equal(lib.get_motion_num_steps().cpu().numpy(), z["num_steps"])
    # the packed records expose the reference's table attributes as views
    for k in ("gts", "grs", "lrs", "gvs", "gavs", "dvs"):
        assert torch.equal(getattr(lib, k).cpu(), tabs[k]), k
    assert torch.equal(lib.length_starts.cpu(), tabs["length_starts"])


@pytest.mark.gpu
@pytest.mark.parametrize("num_motions,n", [(1, 1), (37, 1001), (300, 8192)])
def test_motion_state_matches_oracle_at_size(dev, num_motions, n):
    from pulse_amd.env.motion_lib import MotionLib
    g = syn.make_generator(100 + n)
    tabs = syn.synthetic_motion_library(g, num_motions, 20, 90)
    lib, orc = MotionLib.from_tables(tabs, dev), OracleMotionLib(tabs)
    ids = torch.randint(0, num_motions, (n,), generator=g)
    times = torch.rand(n, generator=g) * tabs["motion_lengths"][ids] * 1.1 - 0.03
    off = torch.randn(n, 3, generator=g)
    want = orc.get_motion_state(ids, times, off)
    res = lib.get_motion_state(ids.to(dev), times.to(dev), off.to(dev))
    _compare(res, want)
    # in-kernel episode clock == the reference's expression (humanoid_im.py:730): (progress + 1) * dt + start + start_offset
    progress = torch.randint(0, 120, (n,), generator=g)
    start, start_off = torch.rand(n, generator=g) * 2.0, torch.rand(n, generator=g) * 0.1
    dt = 1.0 / 30.0
    t_ref = (progress + 1) * dt + start + start_off
    want2 = orc.get_motion_state(ids, t_ref, off)
    res2 = lib.query(ids.to(dev), offset=off.to(dev), progress=progress.to(dev), step_shift=1, dt=dt, start_times=start.to(dev),
                     start_offsets=start_off.to(dev), with_frames=True)
    assert torch.equal(res2["frame_idx0"].cpu() + tabs["length_starts"][ids], want2["frame_idx0"])
    assert torch.equal(res2["blend"].cpu(), want2["blend"])
    assert torch.equal(res2["rg_pos"].cpu(), want2["rg_pos"])
    # outputs can be reused in place
    again = lib.get_motion_state(ids.to(dev), times.to(dev), off.to(dev), out=res)
    assert again["rg_pos"].data_ptr() == res["rg_pos"].data_ptr()


@pytest.mark.gpu
def test_motion_state_rejects_bad_arguments(dev):
    from pulse_amd import _lib
    from pulse_amd.env.motion_lib import MotionLib
    tabs = syn.synthetic_motion_library(syn.make_generator(1), 3, 10, 20)
    lib = MotionLib.from_tables(tabs, dev)
    ids = torch.zeros(4, dtype=torch.int64, device=dev)
    with pytest.raises(TypeError):
        lib.get_motion_state(ids.int(), torch.zeros(4, device=dev))
    with pytest.raises(ValueError):
        lib.get_motion_state(ids, torch.zeros(5, device=dev))
    with pytest.raises(TypeError):
        lib.query(ids)
    assert lib.get_motion_state(ids[:0], torch.zeros(0, device=dev))["rg_pos"].shape == (0, 24, 3)


@pytest.mark.skipif(not os.path.exists("/root/reference/phc/utils/motion_lib_base.py"), reason="reference checkout not mounted")
def test_oracle_time_sampling_matches_reference_methods():
    """sample_time / sample_time_interval / get_motion_length / get_motion_num_steps of the oracle vs MotionLibBase's own method bodies
    (same torch RNG stream in, same times out)."""
    from oracle import refload
    _, tabs = _golden()
    ref = refload.motion_lib_class()()
    ref._motion_lengths, ref._motion_fps, ref._motion_num_frames, ref._device = tabs["motion_lengths"], tabs["motion_fps"], tabs["motion_num_frames"], "cpu"
    orc = OracleMotionLib(tabs)
    ids = torch.randint(0, tabs["motion_lengths"].shape[0], (500,), generator=torch.Generator().manual_seed(2))
    for name in ("sample_time", "sample_time_interval"):
        torch.manual_seed(11)
        want = getattr(ref, name)(ids)
        torch.manual_seed(11)
        got = getattr(orc, name)(ids)
        assert torch.equal(got, want), name
    assert torch.equal(orc.get_motion_length(ids), ref.get_motion_length(ids)) and torch.equal(orc.get_motion_length(), ref.get_motion_length())
    assert torch.equal(orc.get_motion_num_steps(), ref.get_motion_num_steps())
    t = orc.sample_time_interval(ids, generator=torch.Generator().manual_seed(3))
    q = t * 30
    assert torch.allclose(q, q.round(), atol=1e-3)                            # multiples of 1/30 s (humanoid_im.py:652-654)
