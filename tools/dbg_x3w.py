import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
from pulse_amd._lib import ACT_NONE, ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
dev = "cuda:0"
K.F32_MODE = "x3"
def rnd(g, *s): return torch.randn(*s, generator=g)
def padded(t, pitch, fill=float("nan")):
    b = torch.full((t.shape[0], pitch), fill, device=dev); b[:, :t.shape[1]] = t.to(dev); return b
for (m, n, k) in ((200, 300, 70), (512, 512, 64), (256, 256, 16)):
    for epi in (0, EPI_RELU_GRAD):
        g = torch.Generator().manual_seed(n + k)
        dy, w, aux = rnd(g, m, k), rnd(g, k, n) / math.sqrt(k), rnd(g, m, n)
        kp, npad = (k + 3) // 4 * 4, (n + 3) // 4 * 4
        dyd, wd, auxd = padded(dy, kp), padded(w, npad), padded(aux, npad, 1.0)
        res = {}
        for opt in (1, 2):
            K.gemm_set_option(4, opt)
            out = torch.full((m, npad), 5.0, device=dev)
            if epi:
                K.gemm(dyd, wd, out, M=m, N=n, K=k, lda=kp, ldb=npad, ldc=npad, b_layout=GEMM_OUT_CONTIG, epilogue=epi, aux=auxd, ldaux=npad)
            else:
                K.gemm(dyd, wd, out, M=m, N=n, K=k, lda=kp, ldb=npad, ldc=npad, b_layout=GEMM_OUT_CONTIG)
            res[opt] = out.cpu()
        K.gemm_set_option(4, 0)
        acc = dy.double() @ w.double()
        ref = acc * (aux > 0).double() if epi else acc
        for opt in (1, 2):
            e = (res[opt][:, :n].double() - ref).abs()
            bad = (e > 1e-4).nonzero()
            print(f"dx m{m} n{n} k{k} epi{epi} opt{opt}: max err {e.max():.3e}, bad {len(bad)}", "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:40])
# dW
for (m, n, k, split) in ((512, 1024, 4096, 8), (300, 130, 1000, 1)):
    torch.manual_seed(1)
    dy = torch.randn(k, (m + 3) // 4 * 4, device=dev); x = torch.randn(k, (n + 3) // 4 * 4, device=dev)
    res = {}
    for opt in (1, 2):
        K.gemm_set_option(4, opt)
        out = torch.full((split, m * n + m), 7.0, device=dev)
        K.gemm(dy, x, out, M=m, N=n, K=k, lda=dy.shape[1], ldb=x.shape[1], ldc=n, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=split,
               split_stride=m * n + m, rowsum=out, rowsum_off=m * n, stride_rowsum=m)
        res[opt] = out.sum(0).cpu().double()
    K.gemm_set_option(4, 0)
    ref = dy[:, :m].cpu().double().t() @ x[:, :n].cpu().double()
    for opt in (1, 2):
        e = (res[opt][:m * n].view(m, n) - ref).abs()
        bad = (e > 1e-3 * ref.abs().max()).nonzero()
        rs = (res[opt][m * n:] - dy[:, :m].cpu().double().sum(0)).abs().max()
        print(f"dw m{m} n{n} k{k} S{split} opt{opt}: max err {e.max():.3e} bad {len(bad)} rowsum err {rs:.3e}", "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:24])
