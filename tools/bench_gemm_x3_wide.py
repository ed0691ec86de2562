"""A/B of the two x3 tilings (128 x 128: gemm_x3_kernel, 256 x 256: gemm_x3w_kernel) on the cfg2 / cfg3 update shapes, alternating inside one process.

    python tools/bench_gemm_x3_wide.py [--iters 20]

gemm option 4: 1 = never the wide tile, 2 = whenever M, N > 128, 0 = the launcher's own choice.  Activations are ReLU outputs (half zeros), like
the training data; every case reports us per launch and fp32-equivalent TFLOP/s for narrow / wide / automatic."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pulse_amd import kernels as K
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
K.F32_MODE = "x3"


def timeit(fn, iters=args.iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def ab(name, flops, fn, fn_wide=None):
    """fn under option 1 / 2 / 0, twice each, alternating; fn_wide: the wide arm's own launch (e.g. another slab count)."""
    res = {1: [], 2: [], 0: []}
    for _ in range(2):
        for opt in (1, 2, 0):
            K.gemm_set_option(4, opt)
            res[opt].append(timeit(fn_wide if (opt == 2 and fn_wide is not None) else fn))
    K.gemm_set_option(4, 0)
    t = {o: min(v) for o, v in res.items()}
    print(f"{name:46s} narrow {t[1] * 1e6:8.1f} us {flops / t[1] / 1e12:6.1f} TF/s | wide {t[2] * 1e6:8.1f} us {flops / t[2] / 1e12:6.1f} TF/s "
          f"({t[1] / t[2]:.2f} x) | auto {t[0] * 1e6:8.1f} us", flush=True)


def mlp_case(tag, M, in_dim, in_pitch, u, nets):
    """Forward / dX / dW launches of an MLP [u...] over M rows; layer 1 stacked over `nets` networks, upper layers batched."""
    x = torch.zeros(M, in_pitch, device=dev)
    x[:, :in_dim] = torch.randn(M, in_dim, device=dev)
    n1 = nets * u[0]
    w1 = torch.zeros(n1, in_pitch, device=dev)
    w1[:, :in_dim] = torch.randn(n1, in_dim, device=dev) / in_dim ** 0.5
    b1 = torch.randn(n1, device=dev)
    h = [torch.empty(M, nets * uu, device=dev) for uu in u]
    ab(f"{tag} fwd L1 {M}x{n1}x{in_dim}", 2.0 * M * n1 * in_dim,
       lambda: K.gemm(x, w1, h[0], M=M, N=n1, K=in_dim, lda=in_pitch, ldb=in_pitch, ldc=n1, bias=b1, activation=ACT_RELU))
    ws = []
    for l in range(1, len(u)):
        up, uu = u[l - 1], u[l]
        w = torch.randn(nets, uu, up, device=dev) / up ** 0.5
        b = torch.randn(nets, uu, device=dev)
        ws.append(w)
        ab(f"{tag} fwd L{l + 1} {nets}x({M}x{uu}x{up})", 2.0 * nets * M * uu * up,
           lambda: K.gemm(h[l - 1], w, h[l], M=M, N=uu, K=up, lda=nets * up, ldb=up, ldc=nets * uu, bias=b, activation=ACT_RELU, batch=nets,
                          stride_a=up, stride_b=uu * up, stride_c=uu, stride_bias=uu))
    dz = [torch.randn(M, nets * uu, device=dev) * (h[i] > 0) for i, uu in enumerate(u)]
    for l in range(len(u) - 1, 0, -1):
        up, uu = u[l - 1], u[l]
        out = torch.empty(M, nets * up, device=dev)
        ab(f"{tag} dX L{l + 1} {nets}x({M}x{up}x{uu})", 2.0 * nets * M * uu * up,
           lambda: K.gemm(dz[l], ws[l - 1], out, M=M, N=up, K=uu, lda=nets * uu, ldb=up, ldc=nets * up, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD,
                          aux=h[l - 1], ldaux=nets * up, batch=nets, stride_a=uu, stride_b=uu * up, stride_c=up, stride_aux=up))
    # weight gradients: narrow with the slab count the plans use today, wide with the count that fills 256 CUs
    for sn, sw in ((4, 8), (8, 8)):
        slab = n1 * in_pitch
        slabs = torch.empty(max(sn, sw), slab, device=dev)
        mk = lambda S: (lambda: K.gemm(dz[0], x, slabs, M=n1, N=in_pitch, K=M, lda=n1, ldb=in_pitch, ldc=in_pitch, a_layout=GEMM_OUT_CONTIG,
                                       b_layout=GEMM_OUT_CONTIG, split_k=S, split_stride=slab, algo_n=in_dim))
        ab(f"{tag} dW L1 {n1}x{in_pitch}x{M} S={sn}/{sw}", 2.0 * M * n1 * in_dim, mk(sn), mk(sw))
    for l in range(1, len(u)):
        up, uu = u[l - 1], u[l]
        slab = nets * uu * up
        for sn, sw in ((8, 8), (8, 16), (8, 32)):
            slabs = torch.empty(max(sn, sw), slab, device=dev)
            mk = lambda S: (lambda: K.gemm(dz[l], h[l - 1], slabs, M=uu, N=up, K=M, lda=nets * uu, ldb=nets * up, ldc=up, a_layout=GEMM_OUT_CONTIG,
                                           b_layout=GEMM_OUT_CONTIG, batch=nets, stride_a=uu, stride_b=up, stride_c=uu * up, split_k=S, split_stride=slab))
            ab(f"{tag} dW L{l + 1} {nets}x({uu}x{up}x{M}) S={sn}/{sw}", 2.0 * nets * M * uu * up, mk(sn), mk(sw))


mlp_case("cfg2", 16384, 934, 960, [1024, 512], 2)
mlp_case("cfg3", 16384, 3096, 3104, [2048, 1024, 512], 1)
big = torch.randn(4096, 4096, device=dev)
c = torch.empty(4096, 4096, device=dev)
ab("4096^3", 2.0 * 4096 ** 3, lambda: K.gemm(big, big, c, M=4096, N=4096, K=4096, lda=4096, ldb=4096, ldc=4096))
big = torch.randn(8192, 8192, device=dev)
c = torch.empty(8192, 8192, device=dev)
ab("8192^3", 2.0 * 8192 ** 3, lambda: K.gemm(big, big, c, M=8192, N=8192, K=8192, lda=8192, ldb=8192, ldc=8192), )
