"""GPU: the planar fp32-grade GEMM (pulse_gemm_x3p, include/pulse_hip.h 4b) -- operands pre-split into three bf16 planes in HBM,
LDS-DMA staging -- against an fp64 reference, against the in-kernel-split x3 kernel it replaces, and its structural properties
(exact planes, exact small-integer products, tile-position independence, outputs' own planes)."""
import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import ACT_NONE, ACT_RELU, ACT_SILU, EPI_RELU_GRAD, EPI_SILU_GRAD, GEMM_OUT_CONTIG

pytestmark = pytest.mark.gpu


def _mk(m, n, k, dev, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(m, k, generator=g) * scale).to(dev)
    b = (torch.randn(n, k, generator=g) * 0.05).to(dev)
    return a, b


def test_split_planes_is_exact_and_padded(dev):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(37, 69, generator=g) * torch.logspace(-6, 4, 37).unsqueeze(1)).to(dev)
    p = K.split_planes(x)
    assert p.shape == (3, 37, 96) and (p[:, :, 69:] == 0).all()
    assert torch.equal(K.join_planes(p)[:, :69], x)                      # 8 + 8 + 8 significand bits: the split is exact
    # plane 0 is the round-to-nearest-even bf16 of x
    assert torch.equal(p[0, :, :69], x.to(torch.bfloat16).view(torch.int16))
    pt = K.split_planes(x, transpose=True)
    assert pt.shape == (3, 69, 64) and torch.equal(K.join_planes(pt)[:, :37], x.t()) and (pt[:, :, 37:] == 0).all()
    idx = torch.tensor([5, 0, 36, 5], device=dev)
    pg = K.split_planes(x, row_idx=idx)
    assert torch.equal(K.join_planes(pg)[:, :69], x[idx])


@pytest.mark.parametrize("m,n,k", [(256, 128, 32), (300, 200, 100), (130, 69, 934), (4096, 1024, 960), (1024, 1, 512), (64, 2048, 69), (515, 129, 33)])
def test_x3p_matches_fp64_and_the_x3_kernel(dev, m, n, k):
    a, b = _mk(m, n, k, dev, seed=m + n + k)
    bias = torch.randn(n, device=dev)
    pa, pb = K.split_planes(a), K.split_planes(b)
    c = torch.full((m, n + 3), float("nan"), device=dev)[:, :n]
    ldc = c.stride(0)
    # ldc must be a multiple of 4: use a padded pitch
    cbuf = torch.full((m, (n + 3) // 4 * 4 + 4), float("nan"), device=dev)
    K.gemm_x3p(pa, pb, M=m, N=n, K=k, C=cbuf, ldc=cbuf.stride(0), bias=bias)
    got = cbuf[:, :n]
    want = a.double() @ b.double().t() + bias.double()
    ref32 = torch.empty(m, n, device=dev)
    lda = (k + 3) // 4 * 4
    a4 = torch.zeros(m, lda, device=dev); a4[:, :k] = a
    b4 = torch.zeros(n, lda, device=dev); b4[:, :k] = b
    cref = torch.empty(m, (n + 3) // 4 * 4, device=dev)
    K.gemm(a4, b4, cref, M=m, N=n, K=k, lda=lda, ldb=lda, ldc=cref.stride(0), bias=bias, f32_mode="x3")
    err = (got.double() - want).abs().max().item()
    err_x3 = (cref[:, :n].double() - want).abs().max().item()
    scale = want.abs().max().item()
    assert torch.isnan(cbuf[:, n:]).all()                               # nothing written past N
    assert err <= 2e-6 * scale + 1e-6, (err, scale)
    assert err <= 2.0 * err_x3 + 1e-7 * scale, (err, err_x3)            # same arithmetic as the in-kernel split


def test_x3p_small_integer_products_are_exact_and_tile_position_independent(dev):
    g = torch.Generator().manual_seed(3)
    m, n, k = 700, 260, 96
    a = torch.randint(-8, 9, (m, k), generator=g).float().to(dev)
    b = torch.randint(-8, 9, (n, k), generator=g).float().to(dev)
    c = torch.empty(m, 264, device=dev)
    K.gemm_x3p(K.split_planes(a), K.split_planes(b), M=m, N=n, K=k, C=c, ldc=264)
    assert torch.equal(c[:, :n], (a.double() @ b.double().t()).float())
    # a row of A and a row of B give the same output wherever they sit in the tiling
    a2, b2 = a.roll(131, 0), b.roll(77, 0)
    c2 = torch.empty(m, 264, device=dev)
    a3, b3 = torch.randn(m, k, generator=g).to(dev), torch.randn(n, k, generator=g).to(dev)
    c3 = torch.empty(m, 264, device=dev)
    K.gemm_x3p(K.split_planes(a3), K.split_planes(b3), M=m, N=n, K=k, C=c3, ldc=264)
    K.gemm_x3p(K.split_planes(a3.roll(131, 0)), K.split_planes(b3.roll(77, 0)), M=m, N=n, K=k, C=c2, ldc=264)
    assert torch.equal(c2[:, :n], c3[:, :n].roll(131, 0).roll(77, 1))


@pytest.mark.parametrize("act", [ACT_NONE, ACT_RELU, ACT_SILU])
def test_x3p_epilogues_and_output_planes(dev, act):
    m, n, k = 520, 192, 160
    a, b = _mk(m, n, k, dev, seed=9)
    bias = torch.randn(n, device=dev) * 0.1
    pa, pb = K.split_planes(a), K.split_planes(b)
    c = torch.empty(m, n, device=dev)
    z = torch.empty(m, n, device=dev)
    cp = K.alloc_planes(m, n, dev)
    cp.fill_(0x7fc0)                                                     # poison: every element of the valid region must be written
    K.gemm_x3p(pa, pb, M=m, N=n, K=k, C=c, ldc=n, Cp=cp, bias=bias, activation=act, C2=z if act == ACT_SILU else None, ldc2=n)
    pre = a.double() @ b.double().t() + bias.double()
    want = pre if act == ACT_NONE else torch.relu(pre) if act == ACT_RELU else pre * torch.sigmoid(pre)
    assert (c.double() - want).abs().max().item() <= 3e-6 * max(1.0, want.abs().max().item())
    if act == ACT_SILU:
        assert (z.double() - pre).abs().max().item() <= 3e-6 * max(1.0, pre.abs().max().item())
    assert torch.equal(K.join_planes(cp)[:, :n], c)                      # the output's planes are the exact split of the fp32 output
    # gradient epilogues: C = acc * act'(aux)
    dy, w = _mk(m, k, n, dev, seed=11)                                   # dX = dY (m x n) . W (n x k) -> B operand = W^T planes (k x n)
    dy = torch.randn(m, n, device=dev)
    wt = torch.randn(n, k, device=dev) * 0.05
    pdy, pwt = K.split_planes(dy), K.split_planes(wt, transpose=True)
    aux = torch.randn(m, k, device=dev)
    dx = torch.empty(m, k, device=dev)
    if act != ACT_NONE:
        K.gemm_x3p(pdy, pwt, M=m, N=k, K=n, C=dx, ldc=k, epilogue=EPI_RELU_GRAD if act == ACT_RELU else EPI_SILU_GRAD, aux=aux, ldaux=k)
        acc = dy.double() @ wt.double()
        if act == ACT_RELU:
            wantg = acc * (aux > 0)
        else:
            sg = torch.sigmoid(aux.double())
            wantg = acc * (sg * (1 + aux.double() * (1 - sg)))
        assert (dx.double() - wantg).abs().max().item() <= 3e-6 * max(1.0, wantg.abs().max().item())


def test_x3p_batched(dev):
    m, n, k = 384, 128, 64
    g = torch.Generator().manual_seed(5)
    a = torch.randn(m, 2 * k, generator=g).to(dev)                       # two problems side by side in the columns of A ([actor | critic] layout)
    b = torch.randn(2 * n, k, generator=g).to(dev) * 0.1
    pa = K.split_planes(a)                                               # pitch 128: batch z reads columns z * 64 ..
    pb = K.split_planes(b)
    c = torch.empty(m, 2 * n, device=dev)
    bias = torch.randn(2 * n, device=dev)
    K.gemm_x3p(pa, pb, M=m, N=n, K=k, C=c, ldc=2 * n, bias=bias, batch=2, stride_a=k, stride_b=n * pb.stride(1), stride_c=n, stride_bias=n)
    for zb in range(2):
        want = a[:, zb * k:(zb + 1) * k].double() @ b[zb * n:(zb + 1) * n].double().t() + bias[zb * n:(zb + 1) * n].double()
        assert (c[:, zb * n:(zb + 1) * n].double() - want).abs().max().item() <= 3e-6 * want.abs().max().item()


def test_normaliser_writes_the_exact_planes_of_its_output(dev):
    """pulse_rms_normalize_planes: y as before, plus the three bf16 planes of y (pad columns zero) -- bit for bit what pulse_split_planes
    makes of y, with and without the row gather and the moment partials."""
    g = torch.Generator().manual_seed(5)
    rows, cols, pitch = 300, 934, 960
    x = (torch.randn(1000, 936, generator=g) * 3 + 1).to(dev)
    mean = (torch.randn(cols, generator=g, dtype=torch.float64)).to(dev)
    var = (torch.rand(cols, generator=g, dtype=torch.float64) + 0.1).to(dev)
    idx = torch.randperm(1000, generator=g)[:rows].to(dev)
    for row_idx, part in ((None, None), (idx, torch.zeros(16, 2, cols, dtype=torch.float64, device=dev))):
        y0 = torch.full((rows, pitch), float("nan"), device=dev)
        K.rms_normalize(x, mean, var, rows=rows, cols=cols, x_stride=936, y=y0, y_stride=pitch, y_cols=pitch, row_idx=row_idx, moment_partials=part)
        p0 = part.clone() if part is not None else None
        y1 = torch.full((rows, pitch), float("nan"), device=dev)
        planes = torch.full((3, rows, pitch), 0x7fc0, dtype=torch.int16, device=dev)
        K.rms_normalize(x, mean, var, rows=rows, cols=cols, x_stride=936, y=y1, y_stride=pitch, y_cols=pitch, row_idx=row_idx, moment_partials=part, planes=planes)
        assert torch.equal(y0, y1)
        assert torch.equal(planes, K.split_planes(y1))
        assert torch.equal(K.join_planes(planes), y1)
        if part is not None:
            assert torch.equal(part, p0)


def test_actor_critic_layer1_on_the_planar_kernel(dev, monkeypatch):
    """A2CNetwork with layer 1 on the planar GEMM (PULSE_L1_PLANAR=1; the normaliser supplies the planes) against the same network on the
    in-kernel-split kernel: same arithmetic, different accumulation grouping -- fp32-grade agreement; a caller who fills ws['x'] itself
    falls back."""
    from pulse_amd import configs as C
    if K.F32_MODE != "x3":
        pytest.skip("the planar kernel computes the x3 arithmetic (PULSE_GEMM_F32=mfma32 is set)")
    monkeypatch.setenv("PULSE_L1_PLANAR", "1")
    torch.manual_seed(3)
    agent, _ = C.make_agent("cfg1", device="cuda:0", seed=11)
    net = agent.model
    assert net.l1_planar
    agent.init_tensors()
    obs = agent.env_reset()["obs"]
    n = agent.num_actors
    ws = net.workspace(n, train=False)
    net.eval()
    agent._preproc_obs(obs, ws, n)
    assert ws["xp_fresh"] and torch.equal(K.join_planes(ws["xp"])[:, :net.in_pitch], ws["x"])
    net.forward(ws, n)
    assert not ws["xp_fresh"]                                           # consumed
    planar = ws["heads"].clone()
    h_planar = ws["h"][0].clone()
    net.forward(ws, n)                                                  # no fresh planes: the in-kernel-split path on the same ws['x']
    staged = ws["heads"].clone()
    scale = ws["h"][0].abs().max().item()
    assert (h_planar - ws["h"][0]).abs().max().item() <= 4e-6 * scale + 1e-6
    assert (planar - staged).abs().max().item() <= 1e-5 * staged.abs().max().item() + 1e-6
    # the weight planes follow the parameters without anyone being told: poke the flat buffer, run again
    net.flat.mul_(0.5)
    agent._preproc_obs(obs, ws, n)
    net.forward(ws, n)
    half = ws["heads"].clone()
    net.forward(ws, n)
    assert (half - ws["heads"]).abs().max().item() <= 1e-5 * ws["heads"].abs().max().item() + 1e-6
    # critic-only pass
    agent._preproc_obs(obs, ws, n)
    net.eval_critic(ws, n)
    v_planar = ws["val"].clone()
    net.eval_critic(ws, n)
    assert (v_planar - ws["val"]).abs().max().item() <= 1e-5 * ws["val"].abs().max().item() + 1e-6


@pytest.mark.parametrize("rows,m,n,split", [(96, 256, 128, 1), (1000, 300, 200, 1), (16384, 2048, 934, 8), (4099, 515, 129, 4)])
def test_x3p_weight_gradient_form_three_planes(dev, rows, m, n, split):
    """dW(m, n) = sum_r dZ(r, m) X(r, n) over row-major planes (both operands [red][out], transposing LDS reads): fp32-grade."""
    g = torch.Generator().manual_seed(rows + m)
    dz, x = torch.randn(rows, m, generator=g).to(dev), torch.randn(rows, n, generator=g).to(dev)
    ldc = (n + 3) // 4 * 4
    pstride = (m * ldc + 1023) // 1024 * 1024
    slabs = torch.full((split, pstride), float("nan"), device=dev)
    K.gemm_x3p(K.split_planes(dz), K.split_planes(x), M=m, N=n, K=rows, C=slabs, ldc=ldc, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
               split_k=split, split_stride=pstride)
    got = slabs[:, :m * ldc].view(split, m, ldc)[:, :, :n].double().sum(0)
    want = dz.double().t() @ x.double()
    assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item() + 1e-6


@pytest.mark.parametrize("m,n,k", [(300, 200, 100), (16384, 1024, 512), (515, 129, 33)])
def test_x3p_input_gradient_form_three_planes(dev, m, n, k):
    """C(m, n) = sum_k A(m, k) W(k, n), W a [red][out] operand, ReLU mask from the bf16 plane 0 of the activations."""
    g = torch.Generator().manual_seed(m + n)
    a, w, h = torch.randn(m, k, generator=g).to(dev), (torch.randn(k, n, generator=g) * 0.05).to(dev), torch.randn(m, n, generator=g).to(dev)
    ph = K.split_planes(h)
    c = torch.empty(m, (n + 3) // 4 * 4, device=dev)
    K.gemm_x3p(K.split_planes(a), K.split_planes(w), M=m, N=n, K=k, C=c, ldc=c.stride(0), b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD,
               aux=ph[0], ldaux=ph.stride(1))
    want = (a.double() @ w.double()) * (h > 0)                             # sign(h) == sign(bf16(h)) unless h rounds to zero (never here)
    assert (c[:, :n].double() - want).abs().max().item() <= 2e-6 * want.abs().max().item() + 1e-6
