"""GPU: the single-plane mode of pulse_gemm_x3p (include/pulse_hip.h 4b, planes = 1): bf16 operands in HBM, fp32 accumulation, bf16
results -- the three operand layouts (forward, input gradient over [red][out] weights, weight gradient over two [red][out] operands
through the LDS transposing read), their epilogues, split-K, ragged sizes.  A product of bf16 numbers is exact in fp32, so against an
fp64 product of the SAME bf16 operands only the accumulation order differs."""
import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import ACT_NONE, ACT_RELU, ACT_SILU, EPI_RELU_GRAD, EPI_SILU_GRAD, GEMM_OUT_CONTIG, GEMM_RED_CONTIG

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


def _rand(r, c, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(r, c, generator=g) * scale).to(dev)


def test_to_b16_rounds_and_pads(dev):
    x = _rand(37, 69, dev, 1) * torch.logspace(-6, 4, 37, device=dev).unsqueeze(1)
    p = K.to_b16(x)
    assert p.shape == (37, 96) and (p[:, 69:] == 0).all()
    assert torch.equal(p[:, :69], x.to(torch.bfloat16).view(torch.int16))
    pt = K.to_b16(x, transpose=True)
    assert pt.shape == (69, 64) and torch.equal(K.from_b16(pt)[:, :37], _bf(x).t()) and (pt[:, 37:] == 0).all()


# (shapes with M > 8192 or >= 256 tiles of 256 x 128 run on the three-stage ring kernel: ragged M / N, one- and three-k-tile reductions included)
@pytest.mark.parametrize("m,n,k", [(256, 128, 96), (300, 200, 100), (130, 69, 934), (4096, 1024, 1960), (1024, 1, 512), (64, 2048, 69), (515, 129, 33),
                                   (257, 130, 32), (128, 128, 64), (16384, 512, 1024), (8451, 130, 70), (8200, 64, 33), (8200, 136, 32), (9001, 2048, 934),
                                   (12288, 1024, 1960), (8193, 8, 1)])
def test_b16_forward_form(dev, m, n, k):
    a, b = _rand(m, k, dev, m + k), _rand(n, k, dev, n + k, 0.05)
    bias = torch.randn(n, device=dev)
    pa, pb = K.to_b16(a), K.to_b16(b)
    ldc = (n + 3) // 4 * 4 + 4
    c = torch.full((m, ldc), float("nan"), device=dev)
    cp = torch.full((m, K.planes_pitch(n)), 0x7fc0, dtype=torch.int16, device=dev)
    K.gemm_x3p(pa, pb, M=m, N=n, K=k, C=c, ldc=ldc, Cp=cp, bias=bias, planes=1)
    want = _bf(a).double() @ _bf(b).double().t() + bias.double()
    got = c[:, :n]
    assert torch.isnan(c[:, n:]).all()
    scale = want.abs().max().item()
    # the fp32 value before rounding is within accumulation error of fp64; the stored value is that, rounded to bf16
    assert (got.double() - want).abs().max().item() <= 2.0 ** -8 * scale + 1e-6
    assert torch.equal(got, _bf(got))                                   # bf16-representable
    assert torch.equal(K.from_b16(cp)[:, :n], got)                      # Cp is the same matrix as bf16 bits
    n8 = (n + 7) // 8 * 8
    assert (cp[:, n:n8] == 0).all()                                     # pad columns inside the last 8-group are zeros
    # against torch's own bf16 matmul semantics: identical up to accumulation order -> at most one bf16 ulp apart
    ref = (_bf(a) @ _bf(b).t() + bias).to(torch.bfloat16).float()
    assert (got - ref).abs().max().item() <= 2.0 ** -7 * scale


@pytest.mark.parametrize("m,n,k", [(256, 128, 96), (300, 200, 100), (4096, 1960, 1024), (515, 129, 33), (130, 934, 512), (16384, 1024, 512), (8300, 200, 100),
                                   (16384, 512, 69), (12288, 512, 1)])
def test_b16_input_gradient_form(dev, m, n, k):
    """C(m, n) = sum_k A(m, k) W(k, n): W is the forward weight [out = k][in = n], read as a [red][out] operand."""
    a, w = _rand(m, k, dev, m + k), _rand(k, n, dev, n + k, 0.05)
    h = _rand(m, n, dev, 7)
    pa, pw, ph = K.to_b16(a), K.to_b16(w), K.to_b16(h)
    cp = K.alloc_b16(m, n, dev)
    K.gemm_x3p(pa, pw, M=m, N=n, K=k, Cp=cp, planes=1, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=ph, ldaux=ph.stride(0))
    want = (_bf(a).double() @ _bf(w).double()) * (_bf(h) > 0)
    got = K.from_b16(cp)[:, :n]
    scale = want.abs().max().item()
    assert (got.double() - want).abs().max().item() <= 2.0 ** -8 * scale + 1e-6
    assert (cp[:, (n + 7) // 8 * 8:] == 0).all()
    # fp32 aux and fp32 output give the same values
    c = torch.empty(m, (n + 3) // 4 * 4, device=dev)
    h4 = torch.zeros(m, (n + 3) // 4 * 4, device=dev); h4[:, :n] = _bf(h)
    K.gemm_x3p(pa, pw, M=m, N=n, K=k, C=c, ldc=c.stride(0), planes=1, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=h4, ldaux=h4.stride(0))
    assert torch.equal(c[:, :n], got)


@pytest.mark.parametrize("rows,m,n,split", [(96, 256, 128, 1), (1000, 300, 200, 1), (16384, 1024, 934, 8), (4099, 515, 129, 4), (50, 64, 33, 1),
                                             (49152, 1024, 1960, 8), (12288, 1, 512, 8), (5000, 2048, 960, 4), (16384, 2048, 960, 4), (777, 2048, 2048, 2)])
def test_b16_weight_gradient_form(dev, rows, m, n, split):
    """dW(m, n) = sum_r dZ(r, m) X(r, n): both operands row-major over the batch -- the LDS transposing read does the rest."""
    dz, x = _rand(rows, m, dev, rows + m), _rand(rows, n, dev, rows + n)
    pz, px = K.to_b16(dz), K.to_b16(x)
    ldc = (n + 3) // 4 * 4
    pstride = (m * ldc + 1023) // 1024 * 1024
    slabs = torch.full((split, pstride), float("nan"), device=dev)
    K.gemm_x3p(pz, px, M=m, N=n, K=rows, C=slabs, ldc=ldc, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=split, split_stride=pstride)
    got = slabs[:, :m * ldc].view(split, m, ldc)[:, :, :n].double().sum(0)
    want = _bf(dz).double().t() @ _bf(x).double()
    scale = want.abs().max().item()
    tol = (2.0 ** -8 if split == 1 else 1e-5) * scale + 1e-6              # one slab: rounded to bf16; split-K slabs stay fp32
    assert (got - want).abs().max().item() <= tol


def test_b16_small_integers_exact_and_position_independent(dev):
    g = torch.Generator().manual_seed(3)
    rows, m, n = 700, 260, 136
    a = torch.randint(-4, 5, (rows, m), generator=g).float().to(dev)
    b = torch.randint(-4, 5, (rows, n), generator=g).float().to(dev)
    slabs = torch.empty(2, m * 136, device=dev)
    K.gemm_x3p(K.to_b16(a), K.to_b16(b), M=m, N=n, K=rows, C=slabs, ldc=136, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
               split_k=2, split_stride=m * 136)
    assert torch.equal(slabs.view(2, m, 136).sum(0)[:, :n], (a.double().t() @ b.double()).float())
    # forward form: rolled rows give rolled outputs, bit for bit
    x, w = _rand(700, 200, dev, 5), _rand(260, 200, dev, 6)
    c1, c2 = torch.empty(700, 260, device=dev), torch.empty(700, 260, device=dev)
    K.gemm_x3p(K.to_b16(x), K.to_b16(w), M=700, N=260, K=200, C=c1, ldc=260, planes=1)
    K.gemm_x3p(K.to_b16(x.roll(131, 0)), K.to_b16(w.roll(77, 0)), M=700, N=260, K=200, C=c2, ldc=260, planes=1)
    assert torch.equal(c2, c1.roll(131, 0).roll(77, 1))


@pytest.mark.parametrize("act", [ACT_NONE, ACT_RELU, ACT_SILU])
def test_b16_activation_epilogues(dev, act):
    m, n, k = 384, 200, 160
    a, b = _rand(m, k, dev, 11), _rand(n, k, dev, 12, 0.1)
    bias = torch.randn(n, device=dev)
    cp = K.alloc_b16(m, n, dev)
    c2 = torch.empty(m, 200, device=dev)
    K.gemm_x3p(K.to_b16(a), K.to_b16(b), M=m, N=n, K=k, Cp=cp, bias=bias, activation=act, C2=c2 if act == ACT_SILU else None, ldc2=200, planes=1)
    z = (_bf(a).double() @ _bf(b).double().t() + bias.double()).float()
    zb = _bf(z)                                                          # the Linear's bf16 output
    want = zb if act == ACT_NONE else torch.relu(zb) if act == ACT_RELU else _bf(torch.nn.functional.silu(zb))
    got = K.from_b16(cp)[:, :n]
    # rounding boundaries: z sits within accumulation error of the fp64 value, so allow one bf16 ulp
    assert (got - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()
    if act == ACT_SILU:
        assert (c2[:, :n] - zb).abs().max().item() <= 2.0 ** -7 * zb.abs().max().item() and torch.equal(c2[:, :n], _bf(c2[:, :n]))
        # silu-grad epilogue over the stored pre-activation
        dz = K.alloc_b16(m, n, dev)
        up = _rand(m, 96, dev, 13)
        w = _rand(96, n, dev, 14, 0.1)
        K.gemm_x3p(K.to_b16(up), K.to_b16(w), M=m, N=n, K=96, Cp=dz, planes=1, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_SILU_GRAD, aux=c2, ldaux=200)
        pre = c2[:, :n].double()
        sg = torch.sigmoid(pre)
        wantg = _bf((_bf(up).double() @ _bf(w).double()).float()).double() * sg * (1 + pre * (1 - sg))
        assert (K.from_b16(dz)[:, :n].double() - wantg).abs().max().item() <= 2.0 ** -7 * wantg.abs().max().item()


def test_b16_rejects_bad_geometry(dev):
    a, b = K.alloc_b16(64, 64, dev), K.alloc_b16(64, 64, dev)
    c = torch.empty(64, 64, device=dev)
    with pytest.raises(RuntimeError, match="layout combination"):
        K.gemm_x3p(a, b, M=64, N=64, K=64, C=c, ldc=64, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_RED_CONTIG)
    with pytest.raises(RuntimeError, match="planes must be"):
        d, flops, tag = K.make_gemm_x3p_desc(a, b, M=64, N=64, K=64, C=c, ldc=64, planes=1)
        d.planes = 2
        K.launch_gemm_x3p(d, flops, tag)
    with pytest.raises(RuntimeError, match="split-K"):
        K.gemm_x3p(a, b, M=64, N=64, K=64, C=c, ldc=64, planes=1, split_k=2, split_stride=4096)


@pytest.mark.parametrize("m,n,k,batch", [(16384, 1024, 512, 2), (12288, 512, 1, 1), (300, 200, 96, 1), (4096, 1024, 1960, 1)])
def test_b16_output_column_sums(dev, m, n, k, batch):
    """out_colsum: per-row-tile column sums of the output AS STORED (masked, bf16-rounded) -- the bias gradient of the layer whose dZ the
    launch produces.  Summed over the tiles they equal the column sums of the written bf16 matrix."""
    a, b = _rand(m, batch * k, dev, m + k), _rand(batch * n, k, dev, n + k, 0.05)
    h = _rand(m, batch * n, dev, 3)
    pa, pb, ph = K.to_b16(a), K.to_b16(b), K.to_b16(h)
    cp = K.alloc_b16(m, batch * n, dev)
    tiles = K.gemm_x3p_row_tiles(m, n, batch)
    cs = torch.full((tiles + 1, batch * n + 4), float("nan"), device=dev)
    kp = K.planes_pitch(k)
    K.gemm_x3p(pa, pb, M=m, N=n, K=k, Cp=cp, planes=1, batch=batch, stride_a=k if batch > 1 else 0, stride_b=n * kp, stride_cp=n, epilogue=EPI_RELU_GRAD,
               aux=ph, ldaux=ph.stride(0), stride_aux=n, out_colsum=cs, stride_out_colsum=n) if batch == 1 or k % 8 == 0 else pytest.skip("batch stride")
    got = cs[:tiles, :batch * n].double().sum(0)
    want = K.from_b16(cp)[:, :batch * n].double().sum(0)
    assert torch.isnan(cs[tiles]).all() and torch.isnan(cs[:tiles, batch * n:]).all()
    scale = K.from_b16(cp)[:, :batch * n].abs().double().sum(0).max().item() + 1e-9
    assert (got - want).abs().max().item() <= 2e-6 * scale


# ---- the 256 x 256 tile (gemm_b16w_kernel): the launcher picks it only when it does not cost a round of workgroups (the full-size cfg5
# launches above: 16384 x 1024 x 934 with batch 2, the 1024 x 934 / 1024 x 1960 weight gradients); option 3 = 2 forces it on every
# ring-eligible launch with N > 128, so the ragged shapes run through it too.
@pytest.fixture
def wide_tiles():
    K.gemm_set_option(3, 2)
    yield
    K.gemm_set_option(3, 0)


@pytest.mark.parametrize("m,n,k", [(16384, 512, 1024), (8451, 130, 70), (8200, 136, 32), (9001, 2048, 934), (12288, 1024, 1960), (8300, 257, 64), (8193, 384, 65)])
def test_b16_forward_form_wide_tile(dev, wide_tiles, m, n, k):
    test_b16_forward_form(dev, m, n, k)


@pytest.mark.parametrize("m,n,k", [(16384, 1024, 512), (8300, 200, 100), (16384, 512, 69), (12288, 512, 1), (8300, 300, 130)])
def test_b16_input_gradient_form_wide_tile(dev, wide_tiles, m, n, k):
    test_b16_input_gradient_form(dev, m, n, k)


@pytest.mark.parametrize("rows,m,n,split", [(16384, 1024, 934, 8), (49152, 1024, 1960, 8), (5000, 2048, 960, 4), (16384, 2048, 960, 4), (777, 2048, 2048, 2),
                                             (16384, 512, 1024, 16), (9000, 8200, 200, 1), (130, 8200, 136, 1)])
def test_b16_weight_gradient_form_wide_tile(dev, wide_tiles, rows, m, n, split):
    test_b16_weight_gradient_form(dev, rows, m, n, split)


@pytest.mark.parametrize("m,n,k,batch", [(16384, 1024, 512, 2), (12288, 512, 1, 1), (8300, 200, 96, 1)])
def test_b16_output_column_sums_wide_tile(dev, wide_tiles, m, n, k, batch):
    test_b16_output_column_sums(dev, m, n, k, batch)


def test_b16_wide_and_narrow_tiles_agree(dev):
    """Same k order inside a tile either way: the two tilings give bit-identical outputs."""
    m, n, k = 8451, 700, 934
    a, b = _rand(m, k, dev, 1), _rand(n, k, dev, 2, 0.05)
    pa, pb = K.to_b16(a), K.to_b16(b)
    outs = []
    for opt in (1, 2):
        K.gemm_set_option(3, opt)
        try:
            c = torch.empty(m, 704, device=dev)
            K.gemm_x3p(pa, pb, M=m, N=n, K=k, C=c, ldc=704, planes=1)
            outs.append(c[:, :n].clone())
        finally:
            K.gemm_set_option(3, 0)
    assert torch.equal(outs[0], outs[1])
