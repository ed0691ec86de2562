"""The ReLU bit mask of pulse_gemm_f32 (pulse_gemm_desc.relu_mask, ABI v25): a relu forward records one bit per output, a relu-grad launch
with aux = None masks with those bits -- nn.ReLU's backward (phc/learning/network_builder.py:105-124, amp_network_builder.py:230-249) at 1/32
of the traffic of re-reading the activation matrix.

  * the mask IS (h > 0), bit for bit, in the documented layout, whatever tiling wrote it (64-row, 128 x 128, 256 x 256, fp32-MFMA kernel);
  * a relu-grad launch reading the mask gives BIT-IDENTICAL results to the one reading aux = h, across tilings (writer and reader need not
    use the same one), on ragged shapes and batched (actor | critic column halves) launches;
  * the actor / critic network's training pass with masks == the pass without them (PULSE_RELU_BITMASK=0 form), every gradient bit."""
import math

import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG

pytestmark = pytest.mark.gpu


def rnd(g, *shape):
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def unpack(mask, rows, cols):
    """(rows, cols) bool from the documented layout: word [((r >> 6) * 8 + (r & 7)) * ld + (c >> 2)], bit 4 * ((r >> 3) & 7) + (c & 3)."""
    mk = mask.cpu().to(torch.int64) & 0xFFFFFFFF
    r = torch.arange(rows)[:, None]
    c = torch.arange(cols)[None, :]
    w = mk[(r >> 6) * 8 + (r & 7), c >> 2]
    return ((w >> (4 * ((r >> 3) & 7) + (c & 3))) & 1).bool()


@pytest.mark.parametrize("mode", ["x3", "mfma32"])
@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (300, 200, 100), (1024, 512, 96), (129, 132, 17), (64, 128, 32), (2048, 1024, 40), (700, 260, 33)])
def test_forward_mask_and_masked_gradient(dev, monkeypatch, mode, m, n, k):
    monkeypatch.setattr(K, "F32_MODE", mode)
    g = torch.Generator().manual_seed(m + 3 * n + k)
    kp = (k + 3) // 4 * 4
    x = torch.zeros(m, kp); x[:, :k] = rnd(g, m, k)
    w = torch.zeros(n, kp); w[:, :k] = rnd(g, n, k) / math.sqrt(k)
    b = rnd(g, n)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    ldc = (n + 3) // 4 * 4
    k2 = 72
    dy = torch.zeros(m, k2); dy[:, :69] = rnd(g, m, 69)
    w2 = rnd(g, 69, ldc)                                    # [red][out]: the head weights as the dX GEMM reads them
    dyd, w2d = dy.to(dev), w2.to(dev)
    results = []
    for opt_f in ((1, 2) if mode == "x3" else (0,)):         # narrow / wide writer
        K.gemm_set_option(4, opt_f)
        h = torch.full((m, ldc), 7.0, device=dev)
        mask = K.alloc_relu_mask(m, n, dev)
        mask.fill_(-1)                                      # every word the launch owns is rewritten
        K.gemm(xd, wd, h, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=ldc, bias=bd, activation=ACT_RELU, relu_mask=mask, ld_mask=mask.stride(0))
        assert torch.equal(unpack(mask, m, n), (h[:, :n] > 0).cpu()), f"mask != (h > 0), writer tiling {opt_f}"
        for opt_r in ((1, 2) if mode == "x3" else (0,)):     # narrow / wide reader
            K.gemm_set_option(4, opt_r)
            a = torch.full((m, ldc), 5.0, device=dev)
            c = torch.full((m, ldc), 5.0, device=dev)
            K.gemm(dyd, w2d, a, M=m, N=n, K=69, lda=k2, ldb=ldc, ldc=ldc, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=h, ldaux=ldc)
            K.gemm(dyd, w2d, c, M=m, N=n, K=69, lda=k2, ldb=ldc, ldc=ldc, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, relu_mask=mask,
                   ld_mask=mask.stride(0))
            assert torch.equal(a, c), f"masked gradient differs: writer {opt_f}, reader {opt_r}"
            results.append(c)
    K.gemm_set_option(4, 0)
    for r in results[1:]:
        assert torch.equal(results[0], r)
    ref = (dy[:, :69].double() @ w2[:, :n].double()) * (results[0][:, :n].cpu() != 0)      # sanity against fp64 where the mask is set
    live = results[0][:, :n].cpu() != 0
    assert live.float().mean() > 0.2
    err = (results[0][:, :n].cpu().double() - ref)[live].abs().max().item()
    assert err <= 4e-6 * (ref.abs().max().item() + 1e-30)


def test_batched_column_halves(dev, monkeypatch):
    """Actor | critic halves of one (m, 2 u) matrix in ONE batched launch (stride_c = u floats, stride_mask = u / 4 words)."""
    monkeypatch.setattr(K, "F32_MODE", "x3")
    g = torch.Generator().manual_seed(11)
    m, u, k = 640, 384, 128
    x = rnd(g, m, 2 * k).to(dev)
    w = (rnd(g, 2 * u, k) / math.sqrt(k)).to(dev)
    b = rnd(g, 2 * u).to(dev)
    h = torch.zeros(m, 2 * u, device=dev)
    mask = K.alloc_relu_mask(m, 2 * u, dev)
    for opt in (1, 2):
        K.gemm_set_option(4, opt)
        mask.fill_(-1)
        K.gemm(x, w, h, M=m, N=u, K=k, lda=2 * k, ldb=k, ldc=2 * u, bias=b, activation=ACT_RELU, batch=2, stride_a=k, stride_b=u * k, stride_c=u,
               stride_bias=u, relu_mask=mask, ld_mask=mask.stride(0), stride_mask=u // 4)
        assert torch.equal(unpack(mask, m, 2 * u), (h > 0).cpu())
    K.gemm_set_option(4, 0)


def test_network_training_pass_is_bit_identical_with_and_without_masks(dev, monkeypatch):
    from pulse_amd import configs
    from pulse_amd.learning import network as N
    monkeypatch.setattr(K, "F32_MODE", "x3")
    grads = []
    for on in (True, False):
        monkeypatch.setattr(N, "RELU_BITMASK_F32", on)
        torch.manual_seed(5)
        net = N.A2CNetwork(configs.NETWORK_IM, actions_num=69, input_shape=(934,), device=dev)
        m = 1024
        ws = net.workspace(m, True)
        assert ("hmask" in ws) == on
        gsrc = torch.Generator().manual_seed(9)
        ws["x"][:, :934] = rnd(gsrc, m, 934).to(dev)
        net.train()
        net.forward(ws, m)
        ws["dheads"].zero_()
        ws["dmu"].copy_(rnd(gsrc, m, 69).to(dev))
        ws["dval"].copy_(rnd(gsrc, m, 1).to(dev))
        grads.append(net.backward(ws, m).clone())
    assert torch.equal(grads[0], grads[1])
    assert grads[0].abs().sum() > 0


def test_bf16_storage_byte_mask(dev):
    """The same idea in the planar kernels' layout (pulse_gemm_x3p_desc.relu_mask8, ABI v28): a bf16-storage ReLU forward records one byte per row
    and 8 columns, relu-grad launches read it instead of the bf16 activations -- bit-identical outputs, mask == (h > 0), row offsets work."""
    g = torch.Generator().manual_seed(21)
    m, n, k = 1024 + 256, 520, 96
    to16 = lambda t: (t.contiguous().view(torch.int32) + 0x8000 >> 16).to(torch.int16)              # round-half-up is enough for operands here
    x16 = to16(rnd(g, m, k)).to(dev)
    w16 = to16(rnd(g, n, k) / math.sqrt(k)).to(dev)
    npad = (n + 7) // 8 * 8
    h16 = torch.zeros(m, npad, dtype=torch.int16, device=dev)
    mask = K.alloc_relu_mask8(m, n, dev)
    mask.fill_(255)
    K.gemm_x3p(x16, w16, planes=1, M=m, N=n, K=k, Cp=h16, activation=ACT_RELU, relu_mask8=mask)
    h = (h16.to(torch.int32) << 16).view(torch.float32)[:, :n]
    bits = ((mask.cpu().to(torch.int64)[:, :, None] >> torch.arange(8)) & 1).reshape(m, -1)[:, :n].bool()
    assert torch.equal(bits, (h > 0).cpu())
    dy16 = to16(rnd(g, m, 32)).to(dev)
    w2 = to16(rnd(g, 32, npad)).to(dev)                       # [red][out]
    row0 = 256                                               # a row range of the activations (the discriminator's demo rows)
    outs = []
    for use_mask in (False, True):
        z = torch.zeros(m - row0, npad, dtype=torch.int16, device=dev)
        kw = dict(relu_mask8=mask, mask8_off=row0 * mask.stride(0)) if use_mask else dict(aux=h16, ldaux=npad, aux_off=row0 * npad)
        K.gemm_x3p(dy16, w2, planes=1, M=m - row0, N=n, K=32, a_off=row0 * 32, b_layout=GEMM_OUT_CONTIG, Cp=z, epilogue=EPI_RELU_GRAD, **kw)
        outs.append(z)
    assert torch.equal(outs[0], outs[1]) and (outs[0] != 0).any()


def test_mixed_precision_network_pass_is_bit_identical_with_and_without_byte_masks(dev, monkeypatch):
    from pulse_amd import configs
    from pulse_amd.learning import network as N
    grads = []
    for on in (True, False):
        monkeypatch.setattr(N, "RELU_BITMASK", on)
        torch.manual_seed(5)
        net = N.A2CNetwork(configs.NETWORK_IM, actions_num=69, input_shape=(934,), device=dev)
        net.mixed_precision = True
        m = 1024
        ws = net.workspace(m, True)
        assert ws["b16"] and ("hmask8" in ws) == on
        gsrc = torch.Generator().manual_seed(9)
        ws["x"][:, :934] = rnd(gsrc, m, 934).to(dev)
        net.train()
        net.forward(ws, m)
        ws["dheads"].zero_()
        K.to_b16(torch.cat([rnd(gsrc, m, 69), torch.zeros(m, ws["head_pitch16"] - 69), rnd(gsrc, m, 1), torch.zeros(m, ws["head_pitch16"] - 1)], dim=1).to(dev),
                 ws["dheads16"])
        grads.append(net.backward(ws, m).clone())
    assert torch.equal(grads[0], grads[1]) and grads[0].abs().sum() > 0


@pytest.mark.parametrize("tile", [1, 2])                       # gemm option 3: never / always the 256 x 256 bf16-storage tile
@pytest.mark.parametrize("m,n,k,bias", [(512, 256, 64, True), (1000, 520, 96, True), (1280, 1024, 128, False), (300, 136, 64, True), (2048, 2048, 64, True)])
def test_b16_fast_epilogue_rows_equal_the_general_row(dev, tile, m, n, k, bias):
    """[r6] The ReLU-forward and ReLU-gradient launches of the bf16-storage path round a row once (ReLU / the mask select commute with the
    rounding) and read the sign byte and the column sums off the packed words; gemm option 9 sends every row through the general form.
    Outputs, sign bytes and column sums must agree bit for bit -- including rows with NaN, -0, values that round to zero and ragged edges
    (N = 520 / 136: the last 8-column group of a row is partial and takes the general row inside the same launch)."""
    g = torch.Generator().manual_seed(m + n)
    to16 = lambda t: (t.contiguous().view(torch.int32) + 0x8000 >> 16).to(torch.int16)
    x = rnd(g, m, k)
    x[3] = 0.0                                                  # zero rows: outputs are the bias alone (incl. exact zeros where it is zero)
    x[5, 0] = float("nan")
    w = rnd(g, n, k) / math.sqrt(k)
    b = rnd(g, n) * 0.3
    b[::7] = 0.0
    b[1::7] = -0.0
    b[2::7] = 1e-41                                             # rounds to zero in bf16
    x16, w16 = to16(x).to(dev), to16(w).to(dev)
    npad = (n + 7) // 8 * 8
    res = []
    for general in (0, 1):
        K.gemm_set_option(9, general)
        K.gemm_set_option(3, tile)
        try:
            h16 = torch.full((m, npad), 0x7fc0, dtype=torch.int16, device=dev)
            mask = K.alloc_relu_mask8(m, n, dev)
            mask.fill_(255)
            cs = torch.full((K.gemm_x3p_row_tiles(m, n, 1), npad), 9.0, device=dev)
            K.gemm_x3p(x16, w16, planes=1, M=m, N=n, K=k, Cp=h16, activation=ACT_RELU, relu_mask8=mask, bias=b.to(dev) if bias else None, out_colsum=cs)
            lin16 = torch.full((m, npad), 0x7fc0, dtype=torch.int16, device=dev)         # no activation: plain rounding
            K.gemm_x3p(x16, w16, planes=1, M=m, N=n, K=k, Cp=lin16, bias=b.to(dev) if bias else None)
            dy16 = to16(rnd(torch.Generator().manual_seed(9), m, 32)).to(dev)
            w2 = to16(rnd(torch.Generator().manual_seed(10), 32, npad)).to(dev)
            z = torch.full((m, npad), 0x7fc0, dtype=torch.int16, device=dev)
            cs2 = torch.full_like(cs, 9.0)
            K.gemm_x3p(dy16, w2, planes=1, M=m, N=n, K=32, b_layout=GEMM_OUT_CONTIG, Cp=z, epilogue=EPI_RELU_GRAD, relu_mask8=mask, out_colsum=cs2)
        finally:
            K.gemm_set_option(9, 0)
            K.gemm_set_option(3, 0)
        res.append((h16, mask, cs[:, :n], lin16, z, cs2[:, :n]))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)
    h = (res[0][0].to(torch.int32) << 16).view(torch.float32)[:, :n]
    assert (h[3] == 0).any() and (res[0][4] != 0).any() and (not bias or (h[3] > 0).any())
