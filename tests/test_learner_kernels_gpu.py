"""GPU parity of the learner-side kernels through the C ABI.

The GEMM is compared with an fp64 matmul of the same operands (plain PyTorch reference of the
same op); RunningMeanStd / PPO losses / GAE-adjacent pieces are compared with the golden vectors
produced by the real reference code (tests/golden) and with torch autograd for the gradients.
"""
import math

import numpy as np
import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import (ACT_NONE, ACT_RELU, ACT_SILU, EPI_RELU_GRAD, EPI_SILU_GRAD, GEMM_OUT_CONTIG, GEMM_RED_CONTIG)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["x3", "mfma32"])
def f32_mode(request, monkeypatch):
    """Every GEMM test runs on both fp32 paths: the three-way bf16 split on the bf16 MFMA (default) and the fp32 MFMA."""
    monkeypatch.setattr(K, "F32_MODE", request.param)
    return request.param


def rnd(g, *shape):
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def padded(t, pitch, dev):
    """Copy a 2-D tensor into a (rows, pitch) device buffer filled with NaN pads; return the view."""
    buf = torch.full((t.shape[0], pitch), float("nan"), dtype=torch.float32, device=dev)
    buf[:, :t.shape[1]] = t.to(dev)
    return buf


def assert_close64(out, ref64, k):
    out = out.detach().cpu().double()
    scale = ref64.abs().max().item() + 1e-30
    err = (out - ref64).abs().max().item()
    assert err <= 4e-7 * math.sqrt(k) * scale + 1e-6, f"max err {err} (scale {scale}, K={k})"


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (5, 3, 4), (130, 70, 33), (300, 200, 100), (257, 69, 512), (4096, 1, 512), (1024, 512, 934)])
@pytest.mark.parametrize("act", [ACT_NONE, ACT_RELU, ACT_SILU])
def test_gemm_forward(f32_mode, dev, m, n, k, act):
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    x, w, b = rnd(g, m, k), rnd(g, n, k) / math.sqrt(k), rnd(g, n)
    kp = (k + 3) // 4 * 4 + 4
    xd, wd = padded(x, kp, dev), padded(w, kp, dev)
    out = torch.full((m, n + 3), 9.0, device=dev)
    pre = torch.full((m, n + 1), 9.0, device=dev) if act == ACT_SILU else None
    K.gemm(xd, wd, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=n + 3, bias=b.to(dev), activation=act, C2=pre, ldc2=n + 1)
    z = x.double() @ w.double().T + b.double()
    ref = {ACT_NONE: z, ACT_RELU: z.clamp(min=0), ACT_SILU: z * torch.sigmoid(z)}[act]
    assert_close64(out[:, :n], ref, k)
    assert torch.equal(out[:, n:].cpu(), torch.full((m, 3), 9.0))          # nothing written past N
    if pre is not None:
        assert_close64(pre[:, :n], z, k)


def test_gemm_transpose_detecting(f32_mode, dev):
    """A = I with an asymmetric B: catches row/col swaps in the MFMA C/D mapping."""
    n = 192
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) - 100.0
    eye = torch.eye(n)
    out = torch.empty(n, n, device=dev)
    K.gemm(eye.to(dev), b.to(dev), out, M=n, N=n, K=n, lda=n, ldb=n, ldc=n)
    assert torch.equal(out.cpu(), b.T.contiguous())                         # exact: one non-zero product per sum


def test_gemm_batched_shared_input(f32_mode, dev):
    """Two problems in one launch via pointer strides (actor / critic layer pairs)."""
    g = torch.Generator().manual_seed(3)
    m, k, n = 333, 96, 80
    h = rnd(g, m, 2 * k).to(dev)                   # [actor cols | critic cols]
    w = (rnd(g, 2, n, k) / 10).to(dev)
    bias = rnd(g, 2, n).to(dev)
    out = torch.empty(m, 2 * n, device=dev)
    K.gemm(h, w, out, M=m, N=n, K=k, lda=2 * k, ldb=k, ldc=2 * n, bias=bias, activation=ACT_RELU, batch=2,
           stride_a=k, stride_b=n * k, stride_c=n, stride_bias=n)
    for z in range(2):
        ref = (h[:, z * k:(z + 1) * k].cpu().double() @ w[z].cpu().double().T + bias[z].cpu().double()).clamp(min=0)
        assert_close64(out[:, z * n:(z + 1) * n], ref, k)


@pytest.mark.parametrize("m,n,k", [(200, 100, 70), (513, 512, 69), (1000, 512, 1), (2048, 1024, 512)])
@pytest.mark.parametrize("epi", [EPI_RELU_GRAD, EPI_SILU_GRAD])
def test_gemm_dx(f32_mode, dev, m, n, k, epi):
    """dX = (dY W) * act'(aux): A reduction-contiguous, B stored [red][out]."""
    g = torch.Generator().manual_seed(n + k)
    dy, w, aux = rnd(g, m, k), rnd(g, k, n) / math.sqrt(k), rnd(g, m, n)
    kp = (k + 3) // 4 * 4
    out = torch.empty(m, n, device=dev)
    K.gemm(padded(dy, kp, dev), w.to(dev), out, M=m, N=n, K=k, lda=kp, ldb=n, ldc=n, b_layout=GEMM_OUT_CONTIG,
           epilogue=epi, aux=aux.to(dev), ldaux=n)
    acc = dy.double() @ w.double()
    if epi == EPI_RELU_GRAD:
        ref = acc * (aux > 0).double()
    else:
        s = torch.sigmoid(aux.double())
        ref = acc * (s * (1 + aux.double() * (1 - s)))
    assert_close64(out, ref, k)


@pytest.mark.parametrize("m,n,k,split", [(69, 512, 1000, 4), (512, 1024, 4096, 8), (1, 512, 777, 3), (1024, 960, 16384, 16)])
def test_gemm_dw_split_k(f32_mode, dev, m, n, k, split):
    """dW = dY^T X with the batch (reduction) dimension split into slabs + deterministic reduce."""
    g = torch.Generator().manual_seed(k)
    dy, x = rnd(g, k, m), rnd(g, k, n)
    mp, npad = (m + 3) // 4 * 4, (n + 3) // 4 * 4
    count = m * n
    slab = (count + 3) // 4 * 4 + 8
    slabs = torch.full((split, slab), float("nan"), device=dev)
    K.gemm(padded(dy, mp, dev), padded(x, npad, dev), slabs, M=m, N=n, K=k, lda=mp, ldb=npad, ldc=n,
           a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=split, split_stride=slab)
    out = torch.empty(count, device=dev)
    K.reduce_slabs(slabs, split, slab, count, out, scale=0.5)
    ref = 0.5 * (dy.double().T @ x.double())
    assert_close64(out.view(m, n), ref, k)
    # determinism: a second run gives the same bits
    out2 = torch.empty(count, device=dev)
    K.gemm(padded(dy, mp, dev), padded(x, npad, dev), slabs, M=m, N=n, K=k, lda=mp, ldb=npad, ldc=n,
           a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=split, split_stride=slab)
    K.reduce_slabs(slabs, split, slab, count, out2, scale=0.5)
    assert torch.equal(out, out2)


def test_gemm_linearity_full_size(f32_mode, dev):
    """Config-2 layer-1 shape (16384 x 2048 x 960): f(a x) == a f(x), and rows are independent."""
    g = torch.Generator().manual_seed(0)
    m, n, k = 16384, 2048, 960
    x, w = rnd(g, m, k).to(dev), (rnd(g, n, k) / 31).to(dev)
    y1 = K.linear_forward(x, w)
    y2 = K.linear_forward(2 * x, w)
    assert torch.equal(y2, 2 * y1)                                           # exact in binary fp
    sub = K.linear_forward(x[5000:5300].contiguous(), w)
    assert torch.equal(sub, y1[5000:5300])                                   # tile position does not change bits
    ref = x[:64].cpu().double() @ w.cpu().double().T
    assert_close64(y1[:64], ref, k)


@pytest.mark.parametrize("m,n,k", [(512, 512, 4096), (1024, 256, 934)])
def test_gemm_x3_is_fp32_grade(dev, m, n, k):
    """The three-way bf16 split is an fp32 GEMM, not a reduced-precision one: against an fp64 reference its error is no larger than
    the fp32 MFMA's (same operands, wide dynamic range across rows), and far below what ONE bf16 rounding of the operands gives."""
    g = torch.Generator().manual_seed(k)
    x = rnd(g, m, k) * torch.logspace(-6, 4, m).unsqueeze(1)             # rows spanning ten decades
    w = rnd(g, n, k) / math.sqrt(k)
    ref = x.double() @ w.double().T
    scale = ref.abs().amax(dim=1, keepdim=True)                          # per-row scale: small rows must be as accurate as large ones
    errs = {}
    for mode in ("x3", "mfma32"):
        out = torch.empty(m, n, device=dev)
        K.gemm(x.to(dev), w.to(dev), out, M=m, N=n, K=k, lda=k + (-k) % 4, ldb=k + (-k) % 4, ldc=n, f32_mode=mode) if k % 4 == 0 else \
            K.gemm(padded(x, k + (-k) % 4, dev), padded(w, k + (-k) % 4, dev), out, M=m, N=n, K=k, lda=k + (-k) % 4, ldb=k + (-k) % 4, ldc=n,
                   f32_mode=mode)
        e = ((out.cpu().double() - ref).abs() / scale)
        errs[mode] = (e.max().item(), e.pow(2).mean().sqrt().item())
    bf = ((x.bfloat16().double() @ w.bfloat16().double().T) - ref).abs().div(scale).max().item()
    assert errs["x3"][0] <= 1.5 * errs["mfma32"][0] + 1e-9, errs
    assert errs["x3"][1] <= 1.5 * errs["mfma32"][1] + 1e-10, errs
    assert errs["x3"][0] < 1e-3 * bf, (errs, bf)


def test_colsum(dev):
    g = torch.Generator().manual_seed(1)
    m, n, chunks = 5000, 333, 7
    x = rnd(g, m, n)
    ld = 336
    part = torch.zeros(chunks, 400, device=dev)
    K.colsum_partial(padded(x, ld, dev), m, n, ld, chunks, part, 400)
    out = torch.empty(400, device=dev)
    K.reduce_slabs(part, chunks, 400, 400, out)
    np.testing.assert_allclose(out[:n].cpu().numpy(), x.double().sum(0).numpy(), rtol=0, atol=2e-4)


# ------------------------------------------------------------------ RunningMeanStd
def test_running_mean_std_vs_reference_golden(golden, dev):
    """Three train-mode forwards of the real RunningMeanStd (fp64 state), then eval / unnorm."""
    g = golden("rms.npz")
    f = 37
    mean = torch.zeros(f, dtype=torch.float64, device=dev)
    var = torch.ones(f, dtype=torch.float64, device=dev)
    cnt = torch.ones((), dtype=torch.float64, device=dev)
    count = 1.0
    for i in range(3):
        x = g.t(f"x{i}", dev)
        b = x.shape[0]
        y = torch.empty(b, 40, device=dev)
        part = torch.empty(4, 2, f, dtype=torch.float64, device=dev)
        K.rms_normalize(x, mean, var, rows=b, cols=f, x_stride=f, y=y, y_stride=40, y_cols=40, moment_partials=part)
        K.rms_update(mean, var, cnt, part, f, count, b)
        count += b
        np.testing.assert_allclose(y[:, :f].cpu().numpy(), g.np(f"y{i}"), atol=2e-6, rtol=1e-6)
        assert torch.equal(y[:, f:].cpu(), torch.zeros(b, 3))
        np.testing.assert_allclose(mean.cpu().numpy(), g.np(f"mean{i}"), rtol=1e-6, atol=1e-7)   # fp32 batch moments in the reference
        np.testing.assert_allclose(var.cpu().numpy(), g.np(f"var{i}"), rtol=2e-6, atol=1e-7)
        assert cnt.item() == g.np(f"count{i}").item()
    xe = g.t("x_eval", dev)
    y = torch.empty_like(xe)
    K.rms_normalize(xe, mean, var, rows=16, cols=f, x_stride=f, y=y, y_stride=f)
    np.testing.assert_allclose(y.cpu().numpy(), g.np("y_eval"), atol=3e-6, rtol=1e-6)
    z = g.t("z_unnorm_in", dev)
    K.rms_normalize(z, mean, var, rows=16, cols=f, x_stride=f, y=y, y_stride=f, unnorm=True)
    np.testing.assert_allclose(y.cpu().numpy(), g.np("z_unnorm_out"), atol=1e-5, rtol=1e-6)


def test_rms_gather_wide_and_narrow(dev):
    g = torch.Generator().manual_seed(11)
    rows, cols, b = 700, 934, 256
    x = (rnd(g, rows, cols) * 3 + 1).to(dev)
    xs = torch.zeros(rows, 960, device=dev)
    xs[:, :cols] = x
    idx = torch.randperm(rows, generator=g)[:b].to(dev)
    mean = rnd(g, cols).double().to(dev)
    var = (torch.rand(cols, generator=g) + 0.5).double().to(dev)
    y = torch.full((b, 960), 3.0, device=dev)
    part = torch.empty(8, 2, cols, dtype=torch.float64, device=dev)
    K.rms_normalize(xs, mean, var, rows=b, cols=cols, x_stride=960, y=y, y_stride=960, y_cols=960, row_idx=idx, moment_partials=part)
    xg = x[idx]
    ref = ((xg - mean.float()) / torch.sqrt(var.float() + 1e-5)).clamp(-5, 5)
    np.testing.assert_allclose(y[:, :cols].cpu().numpy(), ref.cpu().numpy(), atol=2e-6, rtol=1e-6)
    assert torch.equal(y[:, cols:].cpu(), torch.zeros(b, 26))
    s = part.sum(0)
    np.testing.assert_allclose(s[0].cpu().numpy(), xg.double().sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(s[1].cpu().numpy(), (xg.double() ** 2).sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    # narrow (value normaliser): (B,1)
    v = rnd(g, 5000, 1).to(dev) * 4 + 2
    m1 = torch.tensor([0.3], dtype=torch.float64, device=dev)
    v1 = torch.tensor([2.0], dtype=torch.float64, device=dev)
    out = torch.empty_like(v)
    part1 = torch.empty(16, 2, 1, dtype=torch.float64, device=dev)
    K.rms_normalize(v, m1, v1, rows=5000, cols=1, x_stride=1, y=out, y_stride=1, moment_partials=part1)
    np.testing.assert_allclose(out.cpu().numpy(), ((v - 0.3) / math.sqrt(2.0 + 1e-5)).clamp(-5, 5).cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(part1.sum(0)[0].item(), v.double().sum().item(), rtol=1e-12)


# ------------------------------------------------------------------ policy head, PPO loss, advantages, Adam
def _neglogp(x, mu, logstd):
    sig = torch.exp(logstd)
    return 0.5 * (((x - mu) / sig) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * x.shape[-1] + logstd.sum(-1)


def test_policy_sample(dev):
    g = torch.Generator().manual_seed(5)
    b, a = 301, 69
    mu, noise = rnd(g, b, a), rnd(g, b, a)
    logstd = torch.full((a,), -2.9)
    vraw = rnd(g, b, 1) * 3
    vm, vv = torch.tensor([0.7], dtype=torch.float64), torch.tensor([3.0], dtype=torch.float64)
    mu_d = padded(mu, 72, dev)
    act, sig = torch.empty(b, 72, device=dev), torch.empty(b, 72, device=dev)
    nlp, val = torch.empty(b, device=dev), torch.empty(b, 1, device=dev)
    K.policy_sample(mu_d, 72, logstd.to(dev), noise.to(dev), a, b, a, act, 72, nlp, 1, sigmas=sig, sigmas_stride=72,
                    value_raw=vraw.to(dev), value_stride=1, value_mean=vm.to(dev), value_var=vv.to(dev), values=val, values_stride=1)
    ref_a = mu + torch.exp(logstd) * noise
    np.testing.assert_allclose(act[:, :a].cpu().numpy(), ref_a.numpy(), atol=1e-6)
    np.testing.assert_allclose(nlp.cpu().numpy(), _neglogp(ref_a, mu, logstd).numpy(), rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(sig[:, :a].cpu().numpy(), torch.exp(logstd).expand(b, a).numpy(), rtol=1e-6)
    ref_v = torch.sqrt(vv.float() + 1e-5) * vraw.clamp(-5, 5) + vm.float()
    np.testing.assert_allclose(val.cpu().numpy(), ref_v.numpy(), atol=2e-6)


@pytest.mark.parametrize("clip_value", [False, True])
def test_ppo_loss_and_gradients_vs_autograd(dev, clip_value):
    g = torch.Generator().manual_seed(17)
    n_data, b, a = 900, 512, 69
    e_clip, cc, bc = 0.2, 5.0, 10.0
    logstd = torch.full((a,), -2.9)
    idx = torch.randperm(n_data, generator=g)[:b]
    old_mu = rnd(g, n_data, a) * 0.8
    actions = old_mu + torch.exp(logstd) * rnd(g, n_data, a)
    old_nlp = _neglogp(actions, old_mu, logstd)
    adv = rnd(g, n_data)
    old_val = rnd(g, n_data)
    ret = old_val + 0.5 * rnd(g, n_data)
    mu = (old_mu[idx] + 0.03 * rnd(g, b, a)).requires_grad_(True)        # some ratios leave the clip range
    mu.data[:7] *= 2.0                                                    # |mu| > 1 -> bound loss active
    val = (old_val[idx] + 0.5 * rnd(g, b)).requires_grad_(True)
    # ---- reference: the rl_games / CommonAgent formulas with autograd (common_agent.py:512-520,564-587)
    nlp = _neglogp(actions[idx], mu, logstd)
    ratio = torch.exp(old_nlp[idx] - nlp)
    a_loss = torch.max(-adv[idx] * ratio, -adv[idx] * torch.clamp(ratio, 1 - e_clip, 1 + e_clip))
    if clip_value:
        vpc = old_val[idx] + (val - old_val[idx]).clamp(-e_clip, e_clip)
        c_loss = torch.max((val - ret[idx]) ** 2, (vpc - ret[idx]) ** 2)
    else:
        c_loss = (ret[idx] - val) ** 2
    b_loss = (torch.clamp_max(mu + 1, 0) ** 2 + torch.clamp_min(mu - 1, 0) ** 2).sum(-1)
    loss = a_loss.mean() + cc * c_loss.mean() + bc * b_loss.mean()
    loss.backward()
    sig = torch.exp(logstd)
    kl = (torch.log(sig / sig + 1e-5) + (sig ** 2 + (old_mu[idx] - mu.detach()) ** 2) / (2 * (sig ** 2 + 1e-5)) - 0.5).sum(-1).mean()
    # ---- kernel
    d = lambda t: t.detach().to(dev).contiguous()
    dmu = torch.empty(b, 72, device=dev)
    dval = torch.empty(b, device=dev)
    part = torch.empty(32, 8, device=dev)
    K.ppo_loss(mu=padded(mu.detach(), 72, dev), mu_stride=72, value=d(val), value_stride=1, logstd=d(logstd), old_logstd=d(logstd),
               idx=d(idx), actions=d(actions), actions_stride=a, old_mu=d(old_mu), old_mu_stride=a, old_neglogp=d(old_nlp),
               advantages=d(adv), old_values=d(old_val), returns=d(ret), rows=b, num_actions=a, e_clip=e_clip, critic_coef=cc,
               bounds_loss_coef=bc, clip_value=clip_value, dmu=dmu, dmu_stride=72, dvalue=dval, dvalue_stride=1, partials=part)
    info = part.sum(0).cpu() / b
    np.testing.assert_allclose(info[0].item(), a_loss.mean().item(), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(info[1].item(), c_loss.mean().item(), rtol=1e-5)
    np.testing.assert_allclose(info[2].item(), b_loss.mean().item(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(info[3].item(), (torch.abs(ratio - 1) > e_clip).float().mean().item(), atol=1e-6)
    np.testing.assert_allclose(info[4].item(), kl.item(), rtol=1e-4, atol=1e-6)
    gs = mu.grad.abs().max().item()
    assert (dmu[:, :a].cpu() - mu.grad).abs().max().item() <= 2e-4 * gs
    np.testing.assert_allclose(dval.cpu().numpy(), val.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("a,clip_value,bounds", [(69, False, True), (69, True, False), (153, False, True), (1, True, True), (16, False, False), (160, True, True), (161, False, True)])
def test_ppo_loss_hoisted_form_is_bit_identical_to_plain(dev, a, clip_value, bounds):
    """[r6] The kernel evaluates the column-only terms (sigma, sigma^2, the KL's log and denominator, sum logstd) once per lane; gemm option 8
    selects the plain per-sample form.  Same expressions on the same values: every output must agree bit for bit (161 actions: both calls
    run the plain form -- the hoisted one holds 160 columns)."""
    g = torch.Generator().manual_seed(100 + a)
    n_data, b = 3000, 2050                                       # ragged: the last block is partial, blocks loop over rows
    ap = (a + 3) // 4 * 4
    logstd, old_logstd = rnd(g, a) * 0.3 - 2.0, rnd(g, a) * 0.3 - 2.1
    idx = torch.randperm(n_data, generator=g)[:b]
    old_mu = rnd(g, n_data, a)
    actions = old_mu + torch.exp(old_logstd) * rnd(g, n_data, a)
    mu = old_mu[idx] + 0.05 * rnd(g, b, a)
    mu[:9] *= 2.0
    d = lambda t: t.to(dev).contiguous()
    args = dict(mu=padded(mu, ap + 4, dev), mu_stride=ap + 4, value=d(rnd(g, b)), value_stride=1, logstd=d(logstd), old_logstd=d(old_logstd), idx=d(idx),
                actions=padded(actions, ap, dev), actions_stride=ap, old_mu=padded(old_mu, ap, dev), old_mu_stride=ap,
                old_neglogp=d(_neglogp(actions, old_mu, old_logstd)), advantages=d(rnd(g, n_data)), old_values=d(rnd(g, n_data)), returns=d(rnd(g, n_data)),
                rows=b, num_actions=a, e_clip=0.2, critic_coef=5.0, bounds_loss_coef=10.0 if bounds else None, clip_value=clip_value)
    outs = []
    for plain in (0, 1):
        dmu, dval, part = torch.full((b, ap), 7.0, device=dev), torch.empty(b, device=dev), torch.empty(40, 8, device=dev)
        dmu16, dval16 = torch.zeros(b, ap, dtype=torch.int16, device=dev), torch.zeros(b, 4, dtype=torch.int16, device=dev)
        K.gemm_set_option(8, plain)
        try:
            K.ppo_loss(dmu=dmu, dmu_stride=ap, dvalue=dval, dvalue_stride=1, partials=part, dmu16=dmu16, dmu16_stride=ap, dvalue16=dval16, dvalue16_stride=4, **args)
        finally:
            K.gemm_set_option(8, 0)
        outs.append((dmu, dval, part[:, :5], dmu16, dval16))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert torch.isfinite(outs[0][0]).all() and (outs[0][0][:, a:] == 7.0).all()          # pad columns untouched


def test_losses_vs_reference_golden(golden, dev):
    """_actor_loss / _critic_loss / bound_loss values from the real CommonAgent methods."""
    g = golden("agent_math.npz")
    b, a = 515, 69
    # craft a kernel call whose neglogp equals the golden one: mu = actions => quad = 0, so feed
    # the golden (old - new) difference through old_neglogp instead.
    logstd = torch.zeros(a)
    base = 0.5 * math.log(2 * math.pi) * a
    mu = g.t("loss_mu")
    d = lambda t: t.to(dev).contiguous()
    old_eff = g.t("loss_old_neglogp") - g.t("loss_neglogp") + base          # ratio = exp(old - new) as in the golden
    dmu, dval, part = torch.empty(b, 72, device=dev), torch.empty(b, device=dev), torch.empty(b // 16 + 1, 8, device=dev)
    K.ppo_loss(mu=padded(mu, 72, dev), mu_stride=72, value=d(g.t("loss_values").reshape(-1)), value_stride=1, logstd=d(logstd),
               old_logstd=d(logstd), idx=None, actions=d(mu), actions_stride=a, old_mu=d(mu), old_mu_stride=a, old_neglogp=d(old_eff),
               advantages=d(g.t("loss_adv")), old_values=d(g.t("loss_old_values").reshape(-1)), returns=d(g.t("loss_returns").reshape(-1)),
               rows=b, num_actions=a, e_clip=0.2, critic_coef=1.0, bounds_loss_coef=1.0, clip_value=False, dmu=dmu, dmu_stride=72,
               dvalue=dval, dvalue_stride=1, partials=part)
    info = part.sum(0).cpu().double() / b
    np.testing.assert_allclose(info[0].item(), g.np("actor_loss").astype(np.float64).mean(), rtol=3e-5)
    np.testing.assert_allclose(info[1].item(), g.np("critic_loss").astype(np.float64).mean(), rtol=1e-5)
    np.testing.assert_allclose(info[2].item(), g.np("bound_loss").astype(np.float64).mean(), rtol=1e-5)
    np.testing.assert_allclose(info[3].item(), g.np("actor_clipped").mean(), atol=2.1 / b)   # ratio round-off at the clip edge


def test_advantage_normalize_vs_reference_golden(golden, dev):
    g = golden("agent_math.npz")
    ret, val = g.t("advs_returns", dev).reshape(-1).contiguous(), g.t("advs_values", dev).reshape(-1).contiguous()
    adv = torch.empty_like(ret)
    K.advantage_normalize(ret, val, adv, torch.empty(8, 2, dtype=torch.float64, device=dev))
    np.testing.assert_allclose(adv.cpu().numpy(), g.np("advs_normalized"), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("max_norm", [0.0, 50.0, 0.5])
def test_clip_and_adam_vs_torch(dev, max_norm):
    g = torch.Generator().manual_seed(23)
    n = 100003
    p0 = rnd(g, n)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=2e-5, eps=1e-8)
    p = torch.zeros(n + 1, device=dev)[:n]
    p.copy_(p0)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    part = torch.empty(64, device=dev)
    norm_out = torch.zeros(1, device=dev)
    for step in range(1, 4):
        grad = rnd(g, n) * (0.01 * step)
        p_ref.grad = grad.clone()
        if max_norm > 0:
            ref_norm = torch.nn.utils.clip_grad_norm_([p_ref], max_norm)
        opt.step()
        gd = grad.to(dev)
        K.sqnorm_partial(gd, n, part)
        K.adam_step(p, gd, m, v, n, lr=2e-5, step=step, max_norm=max_norm, sqnorm_partials=part, grad_norm_out=norm_out)
        if max_norm > 0:
            np.testing.assert_allclose(norm_out.item(), ref_norm.item(), rtol=1e-5)          # grad-norm within 1e-4 (north_star)
        np.testing.assert_allclose(p.cpu().numpy(), p_ref.detach().numpy(), atol=2e-7, rtol=1e-6)


def test_adam_step_multi_equals_separate_launches(dev):
    """pulse_adam_step_multi (v21): several flat buffers, one launch, the joint clip -- bit for bit what one pulse_adam_step per buffer gives."""
    g = torch.Generator().manual_seed(31)
    sizes = [300007, 4097, 1]
    def fresh():
        gg = torch.Generator().manual_seed(32)
        return [[rnd(gg, n).to(dev) for _ in range(2)] + [torch.zeros(n, device=dev), torch.zeros(n, device=dev)] for n in sizes]
    a, b = fresh(), fresh()
    part = torch.empty(3 * 64, device=dev)
    na, nb = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    for step in range(1, 4):
        grads = [rnd(g, n).to(dev) * 0.05 for n in sizes]
        for i, (gr, n) in enumerate(zip(grads, sizes)):
            K.sqnorm_partial(gr, n, part[64 * i:64 * (i + 1)])
        kw = dict(lr=3e-4, step=step, weight_decay=1e-3, max_norm=1.0, sqnorm_partials=part)
        for (p, _, m, v), gr, n in zip(a, grads, sizes):
            K.adam_step(p, gr, m, v, n, grad_norm_out=na, **kw)
        K.adam_step_multi([(p, gr, m, v, n) for (p, _, m, v), gr, n in zip(b, grads, sizes)], grad_norm_out=nb, **kw)
        assert torch.equal(na, nb)
        for (pa, _, ma, va), (pb, _, mb, vb) in zip(a, b):
            assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    with pytest.raises(ValueError, match="1..4 groups"):
        K.adam_step_multi([], lr=1e-3, step=1)


def test_rollout_record_matches_reference_sequence(dev):
    """pulse_rollout_record vs the op-by-op sequence of play_steps (amp_agent.py:372-412) with rl_games' AverageMeter."""
    from pulse_amd import kernels as K
    torch.manual_seed(5)
    n, t, slot, max_size = 1000, 7, 3, 100
    cur_r, cur_l = torch.randn(n).abs() * 5, torch.randint(1, 200, (n,)).float()
    meter_r, meter_l = torch.tensor([1.5, 40.0]), torch.tensor([120.0, 40.0])
    vmean, vvar = torch.tensor([0.3], dtype=torch.float64), torch.tensor([2.5], dtype=torch.float64)
    d = {k: v.to(dev) for k, v in dict(cur_r=cur_r.clone(), cur_l=cur_l.clone(), meter_r=meter_r.clone(), meter_l=meter_l.clone()).items()}
    buf_r, buf_nv = torch.zeros(n, t, 1, device=dev), torch.zeros(n, t, 1, device=dev)
    buf_d = torch.zeros(n, t, dtype=torch.uint8, device=dev)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    ref_r, ref_l = [1.5, 40.0], [120.0, 40.0]
    for step in range(4):
        rew = torch.rand(n)
        dones = (torch.rand(n) < (0.0 if step == 2 else 0.3)).long()            # step 2: nobody finishes
        if step == 3:
            dones[:] = 1                                                          # more finished episodes than the window holds
        term = dones * (torch.rand(n) < 0.5).long()
        val = torch.randn(n, 4) * 3
        K.rollout_record(rewards=rew.to(dev), dones=dones.to(dev), terminate=term.to(dev), value_raw=val.to(dev), value_stride=4,
                         value_mean=vmean.to(dev), value_var=vvar.to(dev), value_eps=1e-5, buf_rewards=buf_r[:, slot], buf_next_values=buf_nv[:, slot],
                         buf_dones=buf_d[:, slot], env_stride=t, current_rewards=d["cur_r"], current_lengths=d["cur_l"], meter_rewards=d["meter_r"],
                         meter_lengths=d["meter_l"], meter_max_size=max_size, done_mask=mask)
        # reference sequence on the CPU
        nv = (torch.sqrt(vvar.float() + 1e-5) * torch.clamp(val[:, 0], -5, 5) + vmean.float()) * (1.0 - term.float())
        cur_r = cur_r + rew
        cur_l = cur_l + 1
        idx = dones.nonzero().flatten()
        for st, vals in ((ref_r, cur_r[idx]), (ref_l, cur_l[idx])):
            size = vals.shape[0]
            if size == 0:
                continue
            new_mean = vals.float().mean().item()
            size = float(np.clip(size, 0, max_size))
            old = min(max_size - size, st[1])
            st[0], st[1] = (st[0] * old + new_mean * size) / (old + size), old + size
        nd = 1.0 - dones.float()
        cur_r, cur_l = cur_r * nd, cur_l * nd
        np.testing.assert_array_equal(buf_r[:, slot, 0].cpu().numpy(), rew.numpy())
        np.testing.assert_array_equal(buf_d[:, slot].cpu().numpy(), dones.numpy().astype(np.uint8))
        np.testing.assert_allclose(buf_nv[:, slot, 0].cpu().numpy(), nv.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(d["cur_r"].cpu().numpy(), cur_r.numpy())
        np.testing.assert_array_equal(d["cur_l"].cpu().numpy(), cur_l.numpy())
        np.testing.assert_array_equal(mask.cpu().numpy(), dones.numpy() != 0)
        np.testing.assert_allclose(d["meter_r"].cpu().numpy(), ref_r, rtol=2e-5)
        np.testing.assert_allclose(d["meter_l"].cpu().numpy(), ref_l, rtol=2e-5)
    assert buf_r[:, :slot].abs().sum() == 0 and buf_r[:, slot + 1:].abs().sum() == 0      # only slot `slot` was written


def test_rollout_record_multi_block_with_deferred_meters(dev):
    """meter_partials: several workgroups share the envs and the AverageMeter updates of a whole rollout are applied afterwards, in step
    order, by pulse_rollout_meters -- same buffers, accumulators and meters as the one-workgroup launch that updates them per step."""
    from pulse_amd import kernels as K
    torch.manual_seed(8)
    n, t, steps, max_size, blocks = 5000, 6, 5, 100, 17

    def run(deferred):
        g = torch.Generator().manual_seed(3)
        st = {"cur_r": (torch.rand(n, generator=g) * 5).to(dev), "cur_l": torch.randint(1, 200, (n,), generator=g).float().to(dev),
              "meter_r": torch.tensor([1.5, 40.0], device=dev), "meter_l": torch.tensor([120.0, 40.0], device=dev)}
        buf_r, buf_d = torch.zeros(n, t, 1, device=dev), torch.zeros(n, t, dtype=torch.uint8, device=dev)
        buf_t = torch.zeros(n, t, dtype=torch.uint8, device=dev)
        mask = torch.zeros(n, dtype=torch.bool, device=dev)
        part = torch.full((steps, blocks, 4), float("nan"), device=dev) if deferred else None
        masks = []
        for step in range(steps):
            rew = torch.rand(n, generator=g).to(dev)
            dones = (torch.rand(n, generator=g) < (0.0 if step == 2 else 0.2)).long().to(dev)
            term = (dones * (torch.rand(n, generator=g) < 0.5).long().to(dev))
            K.rollout_record(rewards=rew, dones=dones, terminate=term, value_raw=None, value_stride=0, value_mean=None, value_var=None, value_eps=0.0,
                             buf_rewards=buf_r[:, step], buf_next_values=None, buf_dones=buf_d[:, step], env_stride=t, current_rewards=st["cur_r"],
                             current_lengths=st["cur_l"], meter_rewards=st["meter_r"], meter_lengths=st["meter_l"], meter_max_size=max_size, done_mask=mask,
                             buf_terminate=buf_t[:, step], meter_partials=part[step] if deferred else None)
            masks.append(mask.clone())
        if deferred:
            assert torch.equal(st["meter_r"].cpu(), torch.tensor([1.5, 40.0]))          # untouched until the deferred pass
            K.rollout_meters(part, st["meter_r"], st["meter_l"], max_size)
        return st, buf_r, buf_d, buf_t, masks
    a, b = run(True), run(False)
    for k in ("cur_r", "cur_l"):
        assert torch.equal(a[0][k], b[0][k]), k
    for x, y in zip(a[1:4], b[1:4]):
        assert torch.equal(x, y)
    assert all(torch.equal(x, y) for x, y in zip(a[4], b[4]))
    np.testing.assert_allclose(a[0]["meter_r"].cpu().numpy(), b[0]["meter_r"].cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(a[0]["meter_l"].cpu().numpy(), b[0]["meter_l"].cpu().numpy(), rtol=2e-6)


@pytest.mark.parametrize("m_out,n_out,k_red,split,batch", [(200, 130, 1000, 1, 1), (70, 512, 4096, 8, 2), (1024, 70, 5000, 4, 1)])
def test_gemm_rowsum_is_the_bias_gradient(f32_mode, dev, m_out, n_out, k_red, split, batch):
    """dW pass with pulse_gemm_desc.rowsum: slab sums of the A operand's columns == dY.sum(0) (bias gradient)."""
    torch.manual_seed(m_out + k_red)
    lda, ldb = batch * ((m_out + 3) // 4 * 4), batch * ((n_out + 3) // 4 * 4)
    dy = torch.randn(k_red, lda, device=dev)
    x = torch.randn(k_red, ldb, device=dev)
    cnt = m_out * n_out
    slab = (batch * (cnt + m_out) + 3) // 4 * 4
    out = torch.full((split, slab), 7.0, device=dev)
    K.gemm(dy, x, out, M=m_out, N=n_out, K=k_red, lda=lda, ldb=ldb, ldc=n_out, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
           batch=batch, stride_a=lda // batch, stride_b=ldb // batch, stride_c=cnt, split_k=split, split_stride=slab,
           rowsum=out, rowsum_off=batch * cnt, stride_rowsum=m_out)
    tot = out.sum(0).cpu().double()
    for z in range(batch):
        a = dy[:, z * (lda // batch): z * (lda // batch) + m_out].cpu().double()
        b = x[:, z * (ldb // batch): z * (ldb // batch) + n_out].cpu().double()
        w = tot[z * cnt:(z + 1) * cnt].view(m_out, n_out)
        rs = tot[batch * cnt + z * m_out: batch * cnt + (z + 1) * m_out]
        ref_w, ref_b = a.t() @ b, a.sum(0)
        assert (w - ref_w).abs().max() <= 2e-5 * ref_w.abs().max()
        assert (rs - ref_b).abs().max() <= 2e-5 * max(1.0, ref_b.abs().max())
