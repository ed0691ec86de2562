"""GPU: the glue kernels of the bf16-storage training path (include/pulse_hip.h 4d, csrc/b16_ops.hip + the bf16 outputs of the normaliser,
the PPO loss and the discriminator head) against plain torch on the same inputs; then the two networks trained on bf16 storage against
the SAME networks on the fp32-storage bf16 kernel (identical arithmetic -- operands rounded to bf16, fp32 accumulation, bf16-rounded
outputs -- so the two paths may differ by accumulation order only)."""
import numpy as np
import pytest
import torch

from pulse_amd import configs, kernels as K

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("rows,cols,batch", [(512, 1024, 2), (1024, 1960, 1), (70, 130, 1), (64, 64, 3), (1, 512, 1)])
def test_transpose_to_b16(dev, rows, cols, batch):
    g = torch.Generator().manual_seed(rows + cols)
    ld_in, ld_out = (cols + 3) // 4 * 4 + 4, (rows + 31) // 32 * 32
    x = torch.randn(batch, rows, ld_in, generator=g).to(dev)
    out = torch.full((batch, cols, ld_out), 0x7fc0, dtype=torch.int16, device=dev)
    K.transpose_to_b16(x, out, rows=rows, cols=cols, ld_in=ld_in, ld_out=ld_out, batch=batch, stride_in=rows * ld_in, stride_out=cols * ld_out)
    assert torch.equal(K.from_b16(out[:, :, :rows]), _bf(x[:, :, :cols]).transpose(1, 2))
    assert (out[:, :, rows:] == 0x7fc0).all()                           # columns past rows_in are not touched


@pytest.mark.parametrize("m,n,chunks", [(16384, 2048, 64), (1000, 70, 7), (12288, 1, 64), (256, 192, 1), (33, 1024, 64)])
def test_colsum_partial_b16(dev, m, n, chunks):
    ld = (n + 7) // 8 * 8 + 8
    x = torch.randn(m, ld, device=dev)
    x16 = x.to(torch.bfloat16).view(torch.int16)
    part = torch.full((chunks, n + 3), float("nan"), device=dev)
    K.colsum_partial_b16(x16, m, n, ld, chunks, part, part.stride(0))
    want = _bf(x[:, :n]).double().sum(0)
    got = part[:, :n].double().sum(0)
    assert torch.isnan(part[:, n:]).all()
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-3)


def test_colsum_weighted_b16(dev):
    g = torch.Generator().manual_seed(12)
    m, n, S = 16384, 512, 8
    x = torch.relu(torch.randn(m, n, generator=g)).to(dev)
    w = torch.zeros(m, 32, device=dev)
    w[:, 0] = (torch.randn(m, generator=g) * 1e-3).to(dev)
    x16, w16 = x.to(torch.bfloat16).view(torch.int16), w.to(torch.bfloat16).view(torch.int16)
    slabs = torch.zeros(S, 4096, device=dev)
    lib = __import__("pulse_amd._lib", fromlist=["x"])
    lib.check(lib.load().pulse_colsum_weighted_b16(x16.data_ptr(), m, n, n, w16.data_ptr(), 32, S, slabs.data_ptr() + 4 * 100, slabs.stride(0), None), "colsum_weighted")
    torch.cuda.synchronize()
    want = (_bf(w[:, :1]).double() * _bf(x).double()).sum(0)
    got = slabs[:, 100:100 + n].double().sum(0)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-6)
    assert (slabs[:, :100] == 0).all() and (slabs[:, 100 + n:] == 0).all()


def test_rms_normalize_b16_matches_rounded_fp32(dev):
    from pulse_amd.learning.running_mean_std import RunningMeanStd
    g = torch.Generator().manual_seed(3)
    rows, cols, pitch = 1000, 934, 960
    store = (torch.randn(5000, pitch, generator=g) * 3 + 1).to(dev)
    idx = torch.randint(0, 5000, (rows,), generator=g).to(dev)
    a, b = RunningMeanStd((cols,), device=dev), RunningMeanStd((cols,), device=dev)
    a.running_mean.copy_(torch.randn(cols, dtype=torch.float64, generator=g))
    a.running_var.copy_(torch.rand(cols, dtype=torch.float64, generator=g) + 0.5)
    b.running_mean.copy_(a.running_mean)
    b.running_var.copy_(a.running_var)
    y32 = torch.full((rows, pitch), float("nan"), device=dev)
    y16 = torch.full((rows, pitch), 0x7fc0, dtype=torch.int16, device=dev)
    a.forward(store, row_idx=idx, out=y32, out_cols=pitch)
    b.forward(store, row_idx=idx, out=y16, out_cols=pitch)
    assert torch.equal(K.from_b16(y16), _bf(y32))                       # same values, rounded to nearest even; zero pad columns included
    assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)    # the statistics update is unchanged


def test_disc_penalty_reg_reward(dev):
    g = torch.Generator().manual_seed(9)
    b, k0p = 300, 1984
    G = torch.randn(b, k0p, generator=g).to(dev)
    G[:, 1960:] = 0
    part = torch.zeros(256, device=dev)
    o32 = torch.full((b + 2, k0p), float("nan"), device=dev)
    o16 = torch.zeros(b + 2, k0p, dtype=torch.int16, device=dev)
    K.disc_penalty(G, b, k0p, 0.37, part, out32=o32, out32_off=2 * k0p, ld32=k0p, out16=o16, out16_off=k0p, ld16=k0p)
    np.testing.assert_allclose(part.double().sum().item(), (G.double() ** 2).sum().item(), rtol=1e-6)
    assert torch.equal(o32[2:], G * 0.37) and torch.isnan(o32[:2]).all()
    assert torch.equal(K.from_b16(o16[1:b + 1]), _bf(G * 0.37)) and (o16[0] == 0).all() and (o16[b + 1] == 0).all()
    # regulariser / weight-decay gradients over three ranges of a flat buffer
    flat = torch.randn(5000, generator=g).to(dev)
    grad = torch.randn(5000, generator=g).to(dev)
    g0 = grad.clone()
    rp = torch.zeros(64, 4, device=dev)
    K.disc_reg(flat, grad, [(0, 1000, 0.5), (1200, 2000, 0.0), (4000, 999, -2.0)], rp)
    want = g0.clone()
    want[0:1000] += 0.5 * flat[0:1000]
    want[4000:4999] += -2.0 * flat[4000:4999]
    assert torch.equal(grad, want)
    sq = rp.double().sum(0)
    for r, (o, n) in enumerate(((0, 1000), (1200, 2000), (4000, 999))):
        np.testing.assert_allclose(sq[r].item(), (flat[o:o + n].double() ** 2).sum().item(), rtol=1e-5)
    assert sq[3].item() == 0.0
    # discriminator reward (amp_agent.py:1027-1041), the reference's op sequence
    logits = (torch.randn(1000, 4, generator=g) * 6).to(dev)
    out = torch.empty(1000, 1, device=dev)
    K.disc_reward(logits[:, :1], 1000, 2.0, out)
    prob = 1 / (1 + torch.exp(-logits[:, :1]))
    want = -torch.log(torch.maximum(1 - prob, torch.tensor(0.0001, device=dev))) * 2.0
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_disc_head_b16_matches_fp32_head(dev):
    g = torch.Generator().manual_seed(4)
    b = 257
    L = torch.zeros(4 * b, 4, device=dev)
    L[:, 0] = (torch.randn(4 * b, generator=g) * 2).to(dev)
    d32 = torch.zeros(4 * b, 4, device=dev)
    d16 = torch.zeros(4 * b, 32, dtype=torch.int16, device=dev)
    s32, s16 = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    K.disc_head(L[:3 * b, :1], b, 5.0, d32[:3 * b, :1], s32)
    bg = torch.full((3,), float("nan"), device=dev)
    K.disc_head_b16(L[:3 * b, :1], b, 5.0, d16, s16, bias_grad=bg[1:2])
    assert torch.equal(s32, s16)
    assert torch.equal(K.from_b16(d16[:, 0]), _bf(d32[:, 0])) and (d16[:, 1:] == 0).all() and (d16[3 * b:] == 0).all()
    # v21: the logit bias' gradient = the sum of the logit gradients AS STORED (bf16-rounded), taken by the same launch
    want = K.from_b16(d16[:3 * b, 0]).double().sum().item()
    assert abs(bg[1].item() - want) <= 1e-6 * abs(want) + 1e-9 and torch.isnan(bg[0]) and torch.isnan(bg[2])


def _agent(dev, name, seed, storage):
    import os
    os.environ["PULSE_BF16_STORAGE"] = "1" if storage else "0"
    try:
        torch.manual_seed(seed)
        ag, _ = configs.make_agent(name, device=str(dev), seed=seed, permutation_device="cpu", mixed_precision=True)
        ag.init_tensors()
        ag.obs = ag.env_reset()
        ag._tensors_ready = True
        T, N, A = ag.horizon_length, ag.num_actors, ag.actions_num
        nd = torch.randn(1, T, N, A, generator=torch.Generator().manual_seed(seed)).to(dev)
        ag.noise_provider = lambda e, s: nd[e, s]
        grads = []
        inner = ag._apply_gradients

        def spy(**kw):
            if not grads:
                grads.append((ag.model.grad.clone(), ag.disc.grad.clone()))
            return inner(**kw)
        ag._apply_gradients = spy
        ag.epoch_num = 1
        info = ag.train_epoch()
        # what the workspaces ran on
        ws = ag.model.workspace(ag.minibatch_size, train=True)
        dws = ag.disc.workspace(ag._amp_minibatch_size)
        assert bool(ws["b16"]) == storage and bool(dws["b16"]) == storage
        return ag, info, grads[0]
    finally:
        os.environ.pop("PULSE_BF16_STORAGE", None)


def test_bf16_storage_epoch_matches_fp32_storage_bf16_kernel(dev):
    """cfg5_small, one epoch (24 minibatch steps) from the same weights, noise and minibatches: the bf16-storage networks against the
    fp32-storage bf16 kernel.  First-step gradients of policy and discriminator: identical up to accumulation order (and the bf16
    re-rounding it can flip); the epoch's loss series follow each other."""
    a16, i16, (gp16, gd16) = _agent(dev, "cfg5_small", 31, storage=True)
    a32, i32, (gp32, gd32) = _agent(dev, "cfg5_small", 31, storage=False)
    for name, x, y in (("policy", gp16, gp32), ("disc", gd16, gd32)):
        # the discriminator's flat layouts agree (same ParamBook either way)
        num, den = (x - y).norm().item(), y.norm().item()
        assert num <= 2e-2 * den, (name, num, den)
        assert abs(x.norm().item() - y.norm().item()) <= 5e-3 * den, name
    st = lambda info, key: torch.stack([torch.as_tensor(t).float().reshape(()) for t in info[key]]).cpu().double().numpy()
    for key in ("actor_loss", "critic_loss", "b_loss", "disc_loss", "disc_grad_penalty", "grad_norm", "disc_agent_acc", "disc_demo_acc"):
        u, v = st(i16, key), st(i32, key)
        assert np.abs(u - v)[0] <= 5e-3 * (np.abs(v[0]) + 1e-3), (key, u[0], v[0])
        assert (np.abs(u - v) <= 4e-2 * np.abs(v) + 1e-2).all(), (key, float(np.abs(u - v).max()))     # (the actor loss crosses zero: absolute floor)
    # rollout inference is fp32 on both: identical experience
    for k in ("mus", "values", "actions"):
        assert torch.equal(a16.experience_buffer.tensor_dict[k], a32.experience_buffer.tensor_dict[k]), k


def test_reduce_grads_regions_own_sources_and_partials(dev):
    """pulse_reduce_grads: slab regions with their own slab counts, a region summed from its OWN partial rows (v19), regulariser terms, the
    norm clip's sums of squares -- and the same bits as pulse_reduce_slabs on every region (the data-parallel path reduces with that one)."""
    g = torch.Generator().manual_seed(21)
    S, n = 8, 6000
    slabs = torch.randn(S, n, generator=g).to(dev)
    flat = torch.randn(n, generator=g).to(dev)
    part = torch.randn(37, 1032, generator=g).to(dev)                       # 37 partial rows of a 1024-float range, pitch 1032
    regions = [(0, 2000, 4, 0.25), (2000, 1024, 37, 0.0, part, part.stride(0)), (3024, 976, 8, 0.0), (4000, 2000, 1, -0.5)]
    out = torch.full((n,), float("nan"), device=dev)
    sq, w2 = torch.zeros(1024, device=dev), torch.zeros(1024, 8, device=dev)
    K.ReduceGrads(slabs, n, regions, out, flat=flat).run(scale=0.5, sq_partials=sq, w2_partials=w2)
    want = torch.empty(n, device=dev)
    K.reduce_slabs(slabs, 4, n, 2000, want, scale=0.5)
    K.reduce_slabs(part, 37, part.stride(0), 1024, want, scale=0.5, out_off=2000)
    K.reduce_slabs(slabs, 8, n, 976, want, scale=0.5, slabs_off=3024, out_off=3024)
    K.reduce_slabs(slabs, 1, n, 2000, want, scale=0.5, slabs_off=4000, out_off=4000)
    want[0:2000] += 0.25 * flat[0:2000]
    want[4000:] += -0.5 * flat[4000:]
    assert torch.equal(out, want)
    ref = torch.cat([slabs[:4, :2000].double().sum(0), part[:, :1024].double().sum(0), slabs[:, 3024:4000].double().sum(0), slabs[0, 4000:].double()]) * 0.5
    ref[0:2000] += 0.25 * flat[0:2000].double()
    ref[4000:] += -0.5 * flat[4000:].double()
    assert (out.double() - ref).abs().max().item() <= 1e-5
    np.testing.assert_allclose(sq.double().sum().item(), (out.double() ** 2).sum().item(), rtol=1e-6)
    for r, (o, c) in enumerate(((0, 2000), (2000, 1024), (3024, 976), (4000, 2000))):
        np.testing.assert_allclose(w2.double().sum(0)[r].item(), (flat[o:o + c].double() ** 2).sum().item(), rtol=1e-5)
    with pytest.raises(RuntimeError, match="own source"):
        K.ReduceGrads(slabs, n, [(0, 1024, 3, 0.0, part[:, 1:], part.stride(0))], out).run()


def test_weights_b16_one_launch_equals_refresh_plus_transposes(dev):
    """pulse_weights_to_b16 (v20): the straight bf16 image of a flat parameter buffer and up to four W^T images in one launch, bit for bit
    what pulse_split_planes (one plane) and pulse_transpose_to_b16 write."""
    g = torch.Generator().manual_seed(5)
    count = 1024 * 520 + 77                                                   # not a multiple of 8: the tail piece is zero-filled
    flat = (torch.randn(count + 3, generator=g) * torch.logspace(-4, 3, count + 3)).to(dev)[:count]
    n8 = (count + 7) // 8 * 8
    specs = [dict(x_off=0, rows=1024, cols=512, ld_in=512, ld_out=1024, batch=1, stride_in=0, stride_out=0),
             dict(x_off=4096, rows=130, cols=70, ld_in=72, ld_out=132, batch=3, stride_in=130 * 72, stride_out=70 * 132),
             dict(x_off=1024 * 512, rows=7, cols=513, ld_in=516, ld_out=8, batch=1, stride_in=0, stride_out=0)]
    outs_a = [torch.full((max(1, s["batch"]) * s["cols"], s["ld_out"]), 0x7fc0, dtype=torch.int16, device=dev) for s in specs]
    outs_b = [t.clone() for t in outs_a]
    f16_a = torch.full((n8,), 0x7fc0, dtype=torch.int16, device=dev)
    f16_b = f16_a.clone()
    pa = K.Plan()
    pa.weights_b16(flat, f16_a, count, [dict(x=flat, out=o, **s) for s, o in zip(specs, outs_a)])
    pa.run()
    pb = K.Plan()
    pb.refresh_b16(flat, f16_b, count)
    for s, o in zip(specs, outs_b):
        pb.transpose_b16(flat, o, **s)
    pb.run()
    assert torch.equal(f16_a, f16_b) and (f16_a[count:] == 0).all()
    assert torch.equal(f16_a[:count], flat.to(torch.bfloat16).view(torch.int16))
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    with pytest.raises(ValueError, match="at most four"):
        K.Plan().weights_b16(flat, f16_a, count, [dict(x=flat, out=outs_a[0], **specs[0])] * 5)
