"""GPU: the kernel paths ABI v29 added, against plain torch on seeded random cases.

  * rms_normalize_vec4_kernel in its 1 / 2 / 3 column-group instantiations (rows <= 1024 / <= 2048 / <= 3072 columns): fp32, planes and bf16
    outputs, gathered rows, fp64 moments -- each instantiation must give what the narrow reference formula gives, and the raw-row copy of
    pulse_rms_normalize_copy must be the gathered input bit for bit;
  * pulse_reduce_grads with up to 32 regions and nslabs = 0 regions, against per-region pulse_reduce_slabs (same bits) and fp64;
  * pulse_im_step_args.obs_copy: the second destination equals the first for whole launches, masked subsets (rows outside the subset are
    untouched) and an unaligned destination (the scalar store path).
"""
import numpy as np
import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd import ops
from pulse_amd import synthetic as syn
from pulse_amd._lib import PULSE_IM_REWARD, PULSE_IM_RESET, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS

pytestmark = pytest.mark.gpu


def _bf16_bits(x):
    return x.to(torch.bfloat16).view(torch.int16)


@pytest.mark.parametrize("cols,pitch", [(64, 64), (358, 384), (934, 960), (1024, 1024), (1026, 1056), (1960, 1984), (2048, 2048), (2050, 2080), (3072, 3072)])
def test_normaliser_column_group_variants(dev, cols, pitch):
    g = torch.Generator().manual_seed(cols)
    rows_src, b = 900, 517                                                # ragged: the last workgroup's row range is short
    x = torch.zeros(rows_src, pitch)
    x[:, :cols] = torch.randn(rows_src, cols, generator=g) * 3 + 0.5
    x = x.to(dev)
    idx = torch.randperm(rows_src, generator=g)[:b].to(dev)
    mean = (torch.randn(cols, generator=g) * 0.5).double().to(dev)
    var = (torch.rand(cols, generator=g) + 0.3).double().to(dev)
    xg = x[idx][:, :cols]
    ref = ((xg - mean.float()) / torch.sqrt(var.float() + 1e-5)).clamp(-5, 5)
    part = torch.full((24, 2, cols), float("nan"), dtype=torch.float64, device=dev)
    # fp32 output + raw copy into a strided destination (an experience-buffer slot: row pitch 3 x the row)
    y = torch.full((b, pitch), 7.0, device=dev)
    store = torch.full((b, 3, pitch), -3.0, device=dev)
    raw = store[:, 1]
    assert K.rms_copy_supported(x, cols, y, pitch, raw)
    K.rms_normalize(x, mean, var, rows=b, cols=cols, x_stride=pitch, y=y, y_stride=pitch, y_cols=pitch, row_idx=idx, moment_partials=part, raw_out=raw)
    np.testing.assert_allclose(y[:, :cols].cpu().numpy(), ref.cpu().numpy(), atol=2e-6, rtol=1e-6)
    assert (y[:, cols:] == 0).all()
    c4 = (cols + 3) // 4 * 4
    assert torch.equal(raw[:, :c4], x[idx][:, :c4]) and (raw[:, c4:] == -3.0).all() and (store[:, 0] == -3.0).all() and (store[:, 2] == -3.0).all()
    s = part.sum(0)
    np.testing.assert_allclose(s[0].cpu().numpy(), xg.double().sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(s[1].cpu().numpy(), (xg.double() ** 2).sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    # the same call without the copy gives the same bits
    y2 = torch.full((b, pitch), 7.0, device=dev)
    K.rms_normalize(x, mean, var, rows=b, cols=cols, x_stride=pitch, y=y2, y_stride=pitch, y_cols=pitch, row_idx=idx)
    assert torch.equal(y, y2)
    # bf16 output (the layer-1 operand of the bf16-storage path): the rounding of the fp32 result
    if pitch % 32 == 0:
        y16 = torch.zeros(b, pitch, dtype=torch.int16, device=dev)
        K.rms_normalize(x, mean, var, rows=b, cols=cols, x_stride=pitch, y=y16, y_stride=pitch, y_cols=pitch, row_idx=idx)
        assert torch.equal(y16[:, :cols], _bf16_bits(y[:, :cols])) and (y16[:, cols:] == 0).all()
        # planes: the exact three-way split of y
        planes = torch.zeros(3, b, pitch, dtype=torch.int16, device=dev)
        y3 = torch.empty(b, pitch, device=dev)
        K.rms_normalize(x, mean, var, rows=b, cols=cols, x_stride=pitch, y=y3, y_stride=pitch, y_cols=pitch, row_idx=idx, planes=planes)
        assert torch.equal(y3, y)
        p = planes.view(torch.bfloat16).float()
        assert torch.equal(p[0] + p[1] + p[2], y)                          # three bf16 planes carry the fp32 value exactly


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_reduce_grads_many_regions_and_unwritten_ranges(dev, seed):
    g = torch.Generator().manual_seed(seed)
    S = 8
    nreg = int(torch.randint(9, 33, (1,), generator=g))                    # more than the 8 regions v28 took
    counts = [4 * int(torch.randint(1, 700, (1,), generator=g)) for _ in range(nreg)]
    nsl = [int(torch.randint(0, S + 1, (1,), generator=g)) for _ in range(nreg)]
    nsl[0], nsl[-1] = 0, S
    n = sum(counts)
    slabs = torch.randn(S, n, generator=g).to(dev)
    slabs[:, :counts[0]] = float("nan")                                    # an unwritten range may hold anything: it is never read
    flat = torch.randn(n, generator=g).to(dev)
    alphas = [0.0 if i % 3 else 0.125 for i in range(nreg)]
    regions, off = [], 0
    for c, s_, a in zip(counts, nsl, alphas):
        regions.append((off, c, s_, a))
        off += c
    out = torch.full((n,), float("nan"), device=dev)
    sq, w2 = torch.zeros(512, device=dev), torch.zeros(512, 8, device=dev)
    K.ReduceGrads(slabs, n, regions, out, flat=flat).run(scale=0.25, sq_partials=sq, w2_partials=w2)
    want = torch.zeros(n, device=dev)
    for o, c, s_, a in regions:
        if s_ > 0:
            K.reduce_slabs(slabs, s_, n, c, want, scale=0.25, slabs_off=o, out_off=o)
        want[o:o + c] += a * flat[o:o + c] if a else 0
    assert torch.equal(out, want)
    ref = torch.zeros(n, dtype=torch.float64, device=dev)
    for o, c, s_, a in regions:
        if s_ > 0:
            ref[o:o + c] = slabs[:s_, o:o + c].double().sum(0) * 0.25
        ref[o:o + c] += a * flat[o:o + c].double()
    assert (out.double() - ref).abs().max().item() <= 1e-5
    np.testing.assert_allclose(sq.double().sum().item(), (out.double() ** 2).sum().item(), rtol=1e-6)
    for r, (o, c, _, _) in enumerate(regions[:8]):                         # the regulariser sums cover the first eight regions
        np.testing.assert_allclose(w2.double().sum(0)[r].item(), (flat[o:o + c].double() ** 2).sum().item(), rtol=1e-5)
    with pytest.raises(ValueError):
        K.ReduceGrads(slabs, n, [(0, 4, 1, 0.0)] * 33, out)


@pytest.mark.parametrize("mode", ["all", "mask", "ids", "unaligned"])
def test_im_step_second_observation_destination(dev, mode):
    n = 200
    d = syn.env_step_inputs(syn.make_generator(17), n)
    to = lambda x: x.to(dev)
    rb = to(d["rb"])
    rn, rx = {k: to(v) for k, v in d["ref_now"].items()}, {k: to(v) for k, v in d["ref_next"].items()}
    obs = torch.full((n, 960), 5.0, device=dev)
    store = torch.full((n, 4, 964), -2.0, device=dev)                      # a slot view with a row pitch of 4 x 964 floats
    sink = store[:, 2, 1:961] if mode == "unaligned" else store[:, 2, :960]          # (+ 4 bytes: the scalar store path)
    kw = dict(what=PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS, ref_now=rn, ref_next=rx, dof_force=to(d["dof_force"]),
              dof_vel=to(d["dof_vel"]), progress=to(d["progress"]), pass_time=to(d["pass_time"]), track_ids=list(range(24)), reset_ids=syn.RESET_BODY_IDS,
              term_dist=torch.full((24,), 0.25, device=dev), obs=obs, obs_cols=960, rew=torch.zeros(n, device=dev), rew_raw=torch.zeros(n, 5, device=dev),
              reset=torch.zeros(n, dtype=torch.int64, device=dev), terminate=torch.zeros(n, dtype=torch.int64, device=dev))
    sel = torch.ones(n, dtype=torch.bool)
    if mode == "mask":
        sel = torch.rand(n, generator=torch.Generator().manual_seed(3)) < 0.3
        kw["env_mask"] = sel.to(dev)
    elif mode == "ids":
        sel = torch.zeros(n, dtype=torch.bool)
        ids = torch.randperm(n, generator=torch.Generator().manual_seed(4))[:41]
        sel[ids] = True
        kw["env_ids"] = ids.to(dev)
    ops.im_step(rb, obs_copy=sink, **kw)
    sel = sel.to(dev)
    assert torch.equal(sink[sel], obs[sel]) and (obs[sel] != 5.0).any()
    assert (sink[~sel] == -2.0).all() and (obs[~sel] == 5.0).all()          # rows outside the subset: neither destination is touched
    untouched = store.clone()
    untouched[:, 2, (1 if mode == "unaligned" else 0):(961 if mode == "unaligned" else 960)] = -2.0
    assert (untouched == -2.0).all()                                        # nothing outside the sink's columns was written
    with pytest.raises(ValueError):
        ops.im_step(rb, obs_copy=store[:, 2, :900], **kw)          # narrower than the observation pitch
    # a cached launch re-points the second destination on every call
    cache = {}
    a, b = torch.zeros(n, 960, device=dev), torch.zeros(n, 960, device=dev)
    kw.pop("env_mask", None)
    kw.pop("env_ids", None)
    for dst in (a, b, None):
        ops.im_step(rb, obs_copy=dst, cache=cache, **kw)
    assert "sig" in cache and torch.equal(a, obs) and torch.equal(b, obs)
