"""The 256 x 256 tiling of the x3 fp32 GEMM (pulse_amd/csrc/gemm_x3w.hip) against the 128 x 128 one and against fp64.

Both tilings issue the six plane products of a 16-deep k step in the same order into the same accumulator, so their matrix outputs must
be BIT-IDENTICAL in every form (forward, input gradient, weight gradient with split-K), on full tiles, ragged edges, unaligned output pitches and
k tails; the bias-gradient row sums of the weight-gradient form are summed in a different association and agree to rounding.  gemm option 4: 1 = never the wide tile, 2 = whenever M, N > 128.
Reference of the op: nn.Linear forward / backward, phc/learning/network_builder.py:105-124."""
import math

import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import ACT_NONE, ACT_RELU, ACT_SILU, ACT_SILU_D, EPI_MUL_AUX, EPI_RELU_GRAD, EPI_SILU_GRAD, GEMM_OUT_CONTIG

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def x3(monkeypatch):
    monkeypatch.setattr(K, "F32_MODE", "x3")
    yield
    K.gemm_set_option(4, 0)


def rnd(g, *shape):
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def padded(t, pitch, dev, fill=float("nan")):
    buf = torch.full((t.shape[0], pitch), fill, dtype=torch.float32, device=dev)
    buf[:, :t.shape[1]] = t.to(dev)
    return buf


def both(run):
    """run() under the narrow and the wide tiling -> (narrow outputs, wide outputs)."""
    outs = []
    for opt in (1, 2):
        K.gemm_set_option(4, opt)
        outs.append(run())
    K.gemm_set_option(4, 0)
    torch.cuda.synchronize()
    return outs


def close64(out, ref64, k):
    scale = ref64.abs().max().item() + 1e-30
    err = (out.detach().cpu().double() - ref64).abs().max().item()
    assert err <= 4e-7 * math.sqrt(k) * scale + 1e-6, f"max err {err} (scale {scale}, K={k})"


@pytest.mark.parametrize("m,n,k", [(300, 200, 100), (257, 300, 33), (256, 256, 16), (129, 129, 1), (512, 768, 17), (1024, 512, 934), (2048, 2048, 960),
                                   (700, 257, 515)])
@pytest.mark.parametrize("act,ragged_pitch", [(ACT_NONE, False), (ACT_RELU, False), (ACT_SILU, False), (ACT_RELU, True), (ACT_SILU, True)])
def test_forward_bit_identical(dev, m, n, k, act, ragged_pitch):
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    x, w, b = rnd(g, m, k), rnd(g, n, k) / math.sqrt(k), rnd(g, n)
    kp = (k + 3) // 4 * 4 + 4
    xd, wd, bd = padded(x, kp, dev), padded(w, kp, dev), b.to(dev)
    ldc = n + 3 if ragged_pitch else (n + 3) // 4 * 4                  # n + 3: unaligned rows -> the scalar epilogue

    def run():
        out = torch.full((m, ldc), 9.0, device=dev)
        pre = torch.full((m, ldc), 9.0, device=dev) if act == ACT_SILU else None
        K.gemm(xd, wd, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=ldc, bias=bd, activation=act, C2=pre, ldc2=ldc)
        return out, pre

    (o1, p1), (o2, p2) = both(run)
    assert torch.equal(o1, o2)
    if p1 is not None:
        assert torch.equal(p1, p2)
    z = x.double() @ w.double().T + b.double()
    ref = {ACT_NONE: z, ACT_RELU: z.clamp(min=0), ACT_SILU: z * torch.sigmoid(z)}[act]
    close64(o2[:, :n], ref, k)
    assert torch.equal(o2[:, n:].cpu(), torch.full((m, ldc - n), 9.0))  # nothing written past N


def test_forward_batched_pairs(dev):
    """Two problems in one launch via pointer strides (the actor / critic layer pairs), full tiles and a ragged M."""
    g = torch.Generator().manual_seed(3)
    for m, k, n in ((1024, 512, 256), (777, 96, 384)):
        h = rnd(g, m, 2 * k).to(dev)
        w = (rnd(g, 2, n, k) / 10).to(dev)
        bias = rnd(g, 2, n).to(dev)

        def run():
            out = torch.empty(m, 2 * n, device=dev)
            K.gemm(h, w, out, M=m, N=n, K=k, lda=2 * k, ldb=k, ldc=2 * n, bias=bias, activation=ACT_RELU, batch=2,
                   stride_a=k, stride_b=n * k, stride_c=n, stride_bias=n)
            return out

        o1, o2 = both(run)
        assert torch.equal(o1, o2)
        for z in range(2):
            ref = (h[:, z * k:(z + 1) * k].cpu().double() @ w[z].cpu().double().T + bias[z].cpu().double()).clamp(min=0)
            close64(o2[:, z * n:(z + 1) * n], ref, k)


@pytest.mark.parametrize("m,n,k", [(200, 300, 70), (513, 512, 69), (1000, 512, 1), (2048, 1024, 512), (300, 1030, 40)])
@pytest.mark.parametrize("epi", [EPI_RELU_GRAD, EPI_SILU_GRAD])
def test_dx_bit_identical(dev, m, n, k, epi):
    """dX = (dY W) * act'(aux): A reduction-contiguous, B stored [red][out]."""
    g = torch.Generator().manual_seed(n + k)
    dy, w, aux = rnd(g, m, k), rnd(g, k, n) / math.sqrt(k), rnd(g, m, n)
    kp, npad = (k + 3) // 4 * 4, (n + 3) // 4 * 4
    dyd, wd, auxd = padded(dy, kp, dev), padded(w, npad, dev), padded(aux, npad, dev, fill=1.0)

    def run():
        out = torch.full((m, npad), 5.0, device=dev)
        K.gemm(dyd, wd, out, M=m, N=n, K=k, lda=kp, ldb=npad, ldc=npad, b_layout=GEMM_OUT_CONTIG, epilogue=epi, aux=auxd, ldaux=npad)
        return out

    o1, o2 = both(run)
    assert torch.equal(o1, o2)
    acc = dy.double() @ w.double()
    if epi == EPI_RELU_GRAD:
        ref = acc * (aux > 0).double()
    else:
        s = torch.sigmoid(aux.double())
        ref = acc * (s * (1 + aux.double() * (1 - s)))
    close64(o2[:, :n], ref, k)


@pytest.mark.parametrize("m,n,k,split,batch", [(512, 1024, 4096, 8, 1), (1024, 960, 16384, 16, 1), (2048, 934, 16384, 8, 1), (300, 130, 1000, 1, 1),
                                                (200, 512, 4096, 8, 2), (257, 259, 777, 3, 1), (512, 512, 8192, 32, 2),
                                                (300, 300, 40, 4, 1)])          # (the last one: a split whose k range is empty writes a zero slab)
def test_dw_split_k_and_rowsum_bit_identical(dev, m, n, k, split, batch):
    """dW = dY^T X with the batch (reduction) dimension split into slabs; the A operand's column sums (bias gradient) ride along."""
    torch.manual_seed(m + k)
    lda, ldb = batch * ((m + 3) // 4 * 4), batch * ((n + 3) // 4 * 4)
    dy = torch.randn(k, lda, device=dev)
    x = torch.randn(k, ldb, device=dev)
    cnt = m * n
    slab = (batch * (cnt + m) + 3) // 4 * 4

    def run():
        out = torch.full((split, slab), 7.0, device=dev)
        K.gemm(dy, x, out, M=m, N=n, K=k, lda=lda, ldb=ldb, ldc=n, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
               batch=batch, stride_a=lda // batch, stride_b=ldb // batch, stride_c=cnt, split_k=split, split_stride=slab,
               rowsum=out, rowsum_off=batch * cnt, stride_rowsum=m)
        return out

    o1, o2 = both(run)
    assert torch.equal(o1[:, :batch * cnt], o2[:, :batch * cnt])              # the weight-gradient slabs: bit-identical
    # the row sums ride in a different association (the wide kernel's waves own 4 of every 16 k rows): same value to fp32 rounding of the terms
    r1, r2 = o1[:, batch * cnt:batch * (cnt + m)].double(), o2[:, batch * cnt:batch * (cnt + m)].double()
    assert (r1 - r2).abs().max().item() <= 2e-6 * dy.abs().sum(0).max().item() / split + 1e-6
    assert torch.equal(o1[:, batch * (cnt + m):], o2[:, batch * (cnt + m):])   # nothing written past the row sums
    tot = o2.sum(0).cpu().double()
    for z in range(batch):
        a = dy[:, z * (lda // batch): z * (lda // batch) + m].cpu().double()
        b = x[:, z * (ldb // batch): z * (ldb // batch) + n].cpu().double()
        w = tot[z * cnt:(z + 1) * cnt].view(m, n)
        rs = tot[batch * cnt + z * m: batch * cnt + (z + 1) * m]
        ref_w, ref_b = a.t() @ b, a.sum(0)
        assert (w - ref_w).abs().max() <= 2e-5 * ref_w.abs().max()
        assert (rs - ref_b).abs().max() <= 2e-5 * max(1.0, ref_b.abs().max())


def test_transpose_detecting_and_exact(dev):
    """A = I with an asymmetric B (catches row / column swaps in the wide kernel's C mapping), and small-integer products are exact."""
    n = 384
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) - 100.0
    K.gemm_set_option(4, 2)
    out = torch.empty(n, n, device=dev)
    K.gemm(torch.eye(n).to(dev), b.to(dev), out, M=n, N=n, K=n, lda=n, ldb=n, ldc=n)
    assert torch.equal(out.cpu(), b.T.contiguous())
    g = torch.Generator().manual_seed(1)
    x = torch.randint(-8, 9, (512, 100), generator=g).float()
    w = torch.randint(-8, 9, (300, 100), generator=g).float()
    out = torch.empty(512, 300, device=dev)
    K.gemm(x.to(dev), w.to(dev), out, M=512, N=300, K=100, lda=100, ldb=100, ldc=300)
    assert torch.equal(out.cpu(), x @ w.T)
    K.gemm_set_option(4, 0)


def test_automatic_choice_full_size_matches_narrow(dev):
    """The launcher's own choice (option 0) on the cfg2 update shapes gives the narrow tiling's bits; rows are tile-position independent."""
    g = torch.Generator().manual_seed(0)
    m, n, k = 16384, 2048, 934
    kp = 960
    x = torch.zeros(m, kp)
    x[:, :k] = rnd(g, m, k)
    w = torch.zeros(n, kp)
    w[:, :k] = rnd(g, n, k) / 31
    xd, wd = x.to(dev), w.to(dev)

    def run(opt):
        K.gemm_set_option(4, opt)
        out = torch.empty(m, n, device=dev)
        K.gemm(xd, wd, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=n, activation=ACT_RELU)
        K.gemm_set_option(4, 0)
        return out

    narrow, auto, wide = run(1), run(0), run(2)
    assert torch.equal(narrow, auto) and torch.equal(narrow, wide)
    sub = torch.empty(300, n, device=dev)
    K.gemm_set_option(4, 2)
    K.gemm(xd[5000:5300].contiguous(), wd, sub, M=300, N=n, K=k, lda=kp, ldb=kp, ldc=n, activation=ACT_RELU)
    K.gemm_set_option(4, 0)
    assert torch.equal(sub, wide[5000:5300])
    ref = (x[:64, :k].double() @ w[:, :k].double().T).clamp(min=0)
    close64(wide[:64], ref, k)


@pytest.mark.parametrize("m,n,k", [(512, 512, 64), (300, 257, 70), (2048, 1024, 512)])
@pytest.mark.parametrize("mode", ["x3", "mfma32"])
def test_silu_derivative_activation_and_multiply_epilogue(dev, monkeypatch, m, n, k, mode):
    """ACT_SILU_D: same output as ACT_SILU, C2 = d silu / d z; EPI_MUL_AUX over that C2 == EPI_SILU_GRAD over z, bit for bit, on both tilings
    (the derivative is the expression the SILU_GRAD epilogue evaluates).  Reference: torch.nn.SiLU in the MLPs of amp_network_z_builder.py:341-467."""
    monkeypatch.setattr(K, "F32_MODE", mode)
    g = torch.Generator().manual_seed(m + n + k)
    x, w, b = rnd(g, m, k), rnd(g, n, k) / math.sqrt(k), rnd(g, n)
    dy, w2 = rnd(g, m, k), rnd(g, k, n) / math.sqrt(k)
    kp, npad = (k + 3) // 4 * 4, (n + 3) // 4 * 4
    xd, wd, bd = padded(x, kp, dev), padded(w, kp, dev), b.to(dev)
    dyd, w2d = padded(dy, kp, dev), padded(w2, npad, dev)

    def run():
        res = {}
        for act in (ACT_SILU, ACT_SILU_D):
            out = torch.full((m, npad), 9.0, device=dev)
            c2 = torch.full((m, npad), 9.0, device=dev)
            K.gemm(xd, wd, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=npad, bias=bd, activation=act, C2=c2, ldc2=npad)
            res[act] = (out, c2)
        z, d = res[ACT_SILU][1].clone(), res[ACT_SILU_D][1].clone()
        z[:, n:] = 0.0
        d[:, n:] = 0.0
        g_old = torch.full((m, npad), 5.0, device=dev)
        g_new = torch.full((m, npad), 5.0, device=dev)
        K.gemm(dyd, w2d, g_old, M=m, N=n, K=k, lda=kp, ldb=npad, ldc=npad, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_SILU_GRAD, aux=z, ldaux=npad)
        K.gemm(dyd, w2d, g_new, M=m, N=n, K=k, lda=kp, ldb=npad, ldc=npad, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_MUL_AUX, aux=d, ldaux=npad)
        return res[ACT_SILU][0], res[ACT_SILU_D][0], z, d, g_old, g_new

    for outs in both(run):
        o_s, o_d, z, d, g_old, g_new = outs
        assert torch.equal(o_s, o_d)                                        # the activation itself is unchanged
        assert torch.equal(g_old, g_new)                                    # multiply-by-stored-derivative == recomputed derivative
        zz = z[:, :n].cpu().double()
        sg = torch.sigmoid(zz)
        assert (d[:, :n].cpu().double() - sg * (1 + zz * (1 - sg))).abs().max().item() <= 2e-6
        assert torch.equal(d[:, n:].cpu(), torch.zeros(m, npad - n))
    if mode == "x3":
        a, bb = both(run)
        assert all(torch.equal(u, v) for u, v in zip(a, bb))                # and the two tilings agree bit for bit


def test_random_shapes_both_tilings_agree(dev):
    """Seeded fuzz over ragged shapes, layouts, epilogues, batch strides and split-K: the two tilings agree bit for bit and with fp64."""
    import random
    rng = random.Random(20260926)
    g = torch.Generator().manual_seed(99)
    for case in range(36):
        form = rng.choice(["fwd", "dx", "dw"])
        m, n = rng.randint(129, 900), rng.randint(129, 900)
        k = rng.choice([1, 3, 15, 16, 17, 31, 33, 64, 100, 257, 500])
        batch = rng.choice([1, 1, 2])
        if form == "fwd":
            kp = (k + 3) // 4 * 4 + 4 * rng.randint(0, 2)
            x, w, b = rnd(g, m, batch * kp), rnd(g, batch, n, kp) / math.sqrt(k), rnd(g, batch, n)
            xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
            ldc = batch * n + rng.choice([0, 1, 3, 4])
            act = rng.choice([ACT_NONE, ACT_RELU, ACT_SILU])

            def run():
                out = torch.full((m, ldc), 9.0, device=dev)
                pre = torch.full((m, ldc), 9.0, device=dev) if act == ACT_SILU else None
                K.gemm(xd, wd, out, M=m, N=n, K=k, lda=batch * kp, ldb=kp, ldc=ldc, bias=bd, activation=act, C2=pre, ldc2=ldc, batch=batch, stride_a=kp,
                       stride_b=n * kp, stride_c=n, stride_c2=n, stride_bias=n)
                return out, pre
            (o1, p1), (o2, p2) = both(run)
            assert torch.equal(o1, o2) and (p1 is None or torch.equal(p1, p2)), (case, form, m, n, k, batch, act)
            for z in range(batch):
                zz = x[:, z * kp:z * kp + k].double() @ w[z, :, :k].double().T + b[z].double()
                ref = {ACT_NONE: zz, ACT_RELU: zz.clamp(min=0), ACT_SILU: zz * torch.sigmoid(zz)}[act]
                close64(o2[:, z * n:(z + 1) * n], ref, k)
        elif form == "dx":
            kp, npad = (k + 3) // 4 * 4, (n + 3) // 4 * 4
            dy, w, aux = rnd(g, m, batch * kp), rnd(g, batch, k, npad) / math.sqrt(k), rnd(g, m, batch * npad)
            dyd, wd, auxd = dy.to(dev), w.to(dev), aux.to(dev)
            epi = rng.choice([EPI_RELU_GRAD, EPI_SILU_GRAD, EPI_MUL_AUX])

            def run():
                out = torch.full((m, batch * npad), 5.0, device=dev)
                K.gemm(dyd, wd, out, M=m, N=n, K=k, lda=batch * kp, ldb=npad, ldc=batch * npad, b_layout=GEMM_OUT_CONTIG, epilogue=epi, aux=auxd,
                       ldaux=batch * npad, batch=batch, stride_a=kp, stride_b=k * npad, stride_c=npad, stride_aux=npad)
                return out
            o1, o2 = both(run)
            assert torch.equal(o1, o2), (case, form, m, n, k, batch, epi)
            for z in range(batch):
                acc = dy[:, z * kp:z * kp + k].double() @ w[z, :, :n].double()
                a = aux[:, z * npad:z * npad + n].double()
                sg = torch.sigmoid(a)
                ref = acc * {EPI_RELU_GRAD: (a > 0).double(), EPI_SILU_GRAD: sg * (1 + a * (1 - sg)), EPI_MUL_AUX: a}[epi]
                close64(o2[:, z * npad:z * npad + n], ref, k)
        else:
            kk = rng.choice([40, 777, 2048, 5000])
            split = rng.choice([1, 2, 3, 8])
            lda, ldb = batch * ((m + 3) // 4 * 4), batch * ((n + 3) // 4 * 4)
            dy, x = torch.randn(kk, lda, generator=g).to(dev), torch.randn(kk, ldb, generator=g).to(dev)
            cnt = m * n
            slab = (batch * cnt + 3) // 4 * 4

            def run():
                out = torch.full((split, slab), 7.0, device=dev)
                K.gemm(dy, x, out, M=m, N=n, K=kk, lda=lda, ldb=ldb, ldc=n, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, batch=batch,
                       stride_a=lda // batch, stride_b=ldb // batch, stride_c=cnt, split_k=split, split_stride=slab)
                return out
            o1, o2 = both(run)
            assert torch.equal(o1, o2), (case, form, m, n, kk, batch, split)
            tot = o2.sum(0).cpu().double()
            for z in range(batch):
                ref = dy[:, z * (lda // batch): z * (lda // batch) + m].cpu().double().t() @ x[:, z * (ldb // batch): z * (ldb // batch) + n].cpu().double()
                assert (tot[z * cnt:(z + 1) * cnt].view(m, n) - ref).abs().max() <= 2e-5 * ref.abs().max() + 1e-6


@pytest.mark.parametrize("form", ["fwd_relu_mask", "dx_mul_aux", "fwd_silu_d"])
def test_column_tail_split_is_invisible(dev, form):
    """A narrow column tail that would cost the 256 x 256 tiling a whole extra round of workgroups (cfg3's N = 3096 = 12 x 256 + 24) is launched
    on the 128 x 128 tiling beside the wide main part (gemm_f32.hip; gemm option 5 = 1 switches the split off): same bits, every output
    (C, C2, the ReLU bit mask) and nothing written past N."""
    g = torch.Generator().manual_seed(77)
    m, n, k = 4096, 16 * 256 + 24, 64                       # 16 x 17 = 272 wide tiles = 2 rounds; 16 x 16 = 256 = 1 round + the tail
    x = rnd(g, m, k).to(dev)
    b = rnd(g, n).to(dev)
    ldc = n + 8
    outs = []
    for opt5 in (1, 0):
        K.gemm_set_option(5, opt5)
        K.gemm_set_option(4, 0)
        c = torch.full((m, ldc), 3.0, device=dev)
        c2 = torch.full((m, ldc), 3.0, device=dev)
        if form == "fwd_relu_mask":
            w = (rnd(torch.Generator().manual_seed(1), n, k) / 8).to(dev)
            mask = K.alloc_relu_mask(m, n, dev)
            K.gemm(x, w, c, M=m, N=n, K=k, lda=k, ldb=k, ldc=ldc, bias=b, activation=ACT_RELU, relu_mask=mask, ld_mask=mask.stride(0))
            outs.append((c, mask))
        elif form == "fwd_silu_d":
            w = (rnd(torch.Generator().manual_seed(1), n, k) / 8).to(dev)
            K.gemm(x, w, c, M=m, N=n, K=k, lda=k, ldb=k, ldc=ldc, bias=b, activation=ACT_SILU_D, C2=c2, ldc2=ldc)
            outs.append((c, c2))
        else:
            wt = torch.zeros(k, ldc); wt[:, :n] = rnd(torch.Generator().manual_seed(2), k, n) / 8      # [red][out] weight, as an input gradient reads it
            aux = torch.zeros(m, ldc); aux[:, :n] = rnd(torch.Generator().manual_seed(3), m, n)
            K.gemm(x, wt.to(dev), c, M=m, N=n, K=k, lda=k, ldb=ldc, ldc=ldc, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_MUL_AUX, aux=aux.to(dev), ldaux=ldc)
            outs.append((c,))
        assert torch.equal(c[:, n:].cpu(), torch.full((m, ldc - n), 3.0))
    K.gemm_set_option(5, 0)
    for a, bb in zip(outs[0], outs[1]):
        assert torch.equal(a, bb)
    assert outs[0][0][:, :n].abs().sum() > 0
