"""The skinny-N x3 kernel (pulse_amd/csrc/gemm_x3s.hip: N <= 96 columns over a long M, A reduction-contiguous) against the 128 x 128 / 64 x 128
tiling it replaces for those launches and against fp64.  Same six plane products per k step in the same order into the same accumulator ->
BIT-IDENTICAL outputs (torch.equal), on k tails, ragged M, NaN-filled pitches, both B layouts, batched pairs, bias / ReLU.
gemm option 6: 1 = never the skinny kernel.  Reference of the op: the ``mu`` / ``value`` linears (phc/learning/amp_network_builder.py:127-148),
``z_mu`` / ``z_logvar`` and the decoder's input gradient towards z (phc/learning/amp_network_z_builder.py:341-467)."""
import math

import pytest
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import ACT_NONE, ACT_RELU, GEMM_OUT_CONTIG

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def x3(monkeypatch):
    monkeypatch.setattr(K, "F32_MODE", "x3")
    yield
    K.gemm_set_option(6, 0)


def rnd(g, *shape):
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def padded(t, pitch, dev, fill=float("nan")):
    buf = torch.full((t.shape[0], pitch), fill, dtype=torch.float32, device=dev)
    buf[:, :t.shape[1]] = t.to(dev)
    return buf


def both(run):
    outs = []
    for opt in (1, 0):
        K.gemm_set_option(6, opt)
        outs.append(run())
        tile = K._lib.load().pulse_gemm_last_tile()
        assert (tile == 96) == (opt == 0), f"option 6 = {opt}: served by tile {tile}"
    K.gemm_set_option(6, 0)
    return outs


def close64(out, ref64, k):
    scale = ref64.abs().max().item() + 1e-30
    err = (out.detach().cpu().double() - ref64).abs().max().item()
    assert err <= 4e-7 * math.sqrt(k) * scale + 1e-6, f"max err {err} (scale {scale}, K={k})"


@pytest.mark.parametrize("m,n,k", [(24576, 69, 512), (24576, 1, 512), (24600, 96, 128), (25000, 64, 160), (24576, 32, 3096), (24576, 69, 1024),
                                   (24576, 70, 515), (24576, 33, 16), (24577, 5, 7), (32768, 96, 129)])
@pytest.mark.parametrize("act", [ACT_NONE, ACT_RELU])
def test_forward_bit_identical(dev, m, n, k, act):
    g = torch.Generator().manual_seed(m + 5 * n + k)
    x, w, b = rnd(g, m, k), rnd(g, n, k) / math.sqrt(k), rnd(g, n)
    kp = (k + 3) // 4 * 4 + 4
    xd, wd, bd = padded(x, kp, dev), padded(w, kp, dev), b.to(dev)           # NaN in the pitch padding: the k tail must never read it
    ldc = n + 3

    def run():
        out = torch.full((m, ldc), 9.0, device=dev)
        K.gemm(xd, wd, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=ldc, bias=bd, activation=act)
        return out

    o1, o2 = both(run)
    assert torch.equal(o1, o2)
    z = x.double() @ w.double().T + b.double()
    close64(o2[:, :n], z.clamp(min=0) if act == ACT_RELU else z, k)
    assert torch.equal(o2[:, n:].cpu(), torch.full((m, ldc - n), 9.0))      # nothing written past N


@pytest.mark.parametrize("m,n,k", [(24576, 32, 3096), (24576, 69, 512), (24580, 96, 300), (24576, 8, 77)])
def test_out_contiguous_b_bit_identical(dev, m, n, k):
    """B stored [red][out] (the weight as an input-gradient launch reads it: dz = dh . W[:, z columns])."""
    g = torch.Generator().manual_seed(m + n + k)
    x = rnd(g, m, k)
    ldb = (n + 3) // 4 * 4 + 40                                            # the operand is a column range of a wider matrix
    wt = rnd(g, k, ldb) / math.sqrt(k)
    kp = (k + 3) // 4 * 4
    xd, wd = padded(x, kp, dev, 0.0), wt.to(dev)

    def run():
        out = torch.full((m, n), 9.0, device=dev)
        K.gemm(xd, wd, out, M=m, N=n, K=k, lda=kp, ldb=ldb, ldc=n, b_layout=GEMM_OUT_CONTIG)
        return out

    o1, o2 = both(run)
    assert torch.equal(o1, o2)
    close64(o2, x.double() @ wt[:, :n].double(), k)


def test_batched_heads_of_the_actor_critic_pair(dev):
    """The cfg2 head launch: actor | critic halves of h (m, 2 u) against two (A, u) weight blocks, outputs at columns 0 and a_pitch."""
    g = torch.Generator().manual_seed(4)
    m, u, a, ap = 16384, 512, 69, 72
    h = rnd(g, m, 2 * u).clamp(min=0).to(dev)
    w = (rnd(g, 2 * a, u) / math.sqrt(u)).to(dev)
    w[a + 1:] = 0                                                          # the value head: one real row
    b = rnd(g, 2 * ap).to(dev)

    def run():
        out = torch.zeros(m, 2 * ap, device=dev)
        K.gemm(h, w, out, M=m, N=a, K=u, lda=2 * u, ldb=u, ldc=2 * ap, bias=b, batch=2, stride_a=u, stride_b=a * u, stride_c=ap, stride_bias=ap)
        return out

    o1, o2 = both(run)
    assert torch.equal(o1, o2)
    ref = h[:, :u].double().cpu() @ w[:a].double().cpu().T + b[:a].double().cpu()
    close64(o2[:, :a], ref, u)


def test_small_launches_keep_the_old_tiling(dev):
    """Below 192 row tiles the 128-row workgroups would leave the chip idle, above 256 they need a second round: the launcher keeps the
    64 x 128 tile (the rollout's M = 4096 heads; a 65536-row launch)."""
    w = torch.randn(69, 512, device=dev)
    for m in (4096, 65536):
        x = torch.randn(m, 512, device=dev)
        out = torch.zeros(m, 72, device=dev)
        K.gemm(x, w, out, M=m, N=69, K=512, lda=512, ldb=512, ldc=72)
        assert K._lib.load().pulse_gemm_last_tile() != 96
