"""Time the planar GEMM (pulse_gemm_x3p) against the in-kernel-split x3 kernel on the cfg2 forward / dX shapes (run on the GPU box):
    python tools/bench_gemm_x3p.py [--reps 30]
ReLU-like activations (half zeros) as the A operand of the upper layers, random weights; every case is timed after a warm-up of the
same kernel (the chip's clock depends on what ran just before)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulse_amd import kernels as K  # noqa: E402
from pulse_amd._lib import ACT_RELU  # noqa: E402


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dev = "cuda:0"
    cases = [("L1 fwd  16384x2048x934", 16384, 2048, 934, 1), ("L2 fwd  16384x512x1024 x2", 16384, 512, 1024, 2), ("dX L2   16384x1024x512 x2", 16384, 1024, 512, 2),
             ("head    16384x69x512 x2", 16384, 69, 512, 2), ("rollout L1 4096x2048x934", 4096, 2048, 934, 1), ("rollout L2 4096x512x1024 x2", 4096, 512, 1024, 2),
             ("square  4096^3", 4096, 4096, 4096, 1), ("square  8192^3", 8192, 8192, 8192, 1)]
    print(f"{'case':32s} {'x3 us':>9s} {'TF/s':>7s} {'x3p us':>9s} {'TF/s':>7s} {'x3p+planes us':>14s} {'TF/s':>7s}")
    for name, m, n, k, batch in cases:
        kp = (k + 31) // 32 * 32
        x = torch.relu(torch.randn(m, batch * kp, device=dev))
        x[:, k:kp] = 0
        w = torch.randn(batch * n, kp, device=dev) * 0.03
        w[:, k:] = 0
        bias = torch.randn(batch * n, device=dev)
        nn = (n + 3) // 4 * 4
        c = torch.empty(m, batch * nn, device=dev)
        flops = 2.0 * m * n * k * batch
        f_x3 = lambda: K.gemm(x, w, c, M=m, N=n, K=k, lda=batch * kp, ldb=kp, ldc=batch * nn, bias=bias, activation=ACT_RELU, batch=batch, stride_a=kp,
                              stride_b=n * kp, stride_c=nn, stride_bias=n, f32_mode="x3")
        t_x3 = timed(f_x3, a.reps)
        px, pw = K.split_planes(x), K.split_planes(w)
        cp = K.alloc_planes(m, batch * ((n + 31) // 32 * 32), dev)
        kw = dict(M=m, N=n, K=k, bias=bias, activation=ACT_RELU, batch=batch, stride_a=kp, stride_b=n * pw.stride(1), stride_c=nn, stride_bias=n)
        f_p = lambda: K.gemm_x3p(px, pw, C=c, ldc=batch * nn, **kw)
        t_p = timed(f_p, a.reps)
        f_pp = lambda: K.gemm_x3p(px, pw, C=c, ldc=batch * nn, Cp=cp, stride_cp=(n + 31) // 32 * 32, **kw)
        t_pp = timed(f_pp, a.reps)
        print(f"{name:32s} {t_x3:9.1f} {flops / t_x3 * 1e-6:7.1f} {t_p:9.1f} {flops / t_p * 1e-6:7.1f} {t_pp:14.1f} {flops / t_pp * 1e-6:7.1f}", flush=True)


def backward_forms(reps):
    """The cfg2 backward shapes: x3 kernel (fp32 operands) vs the planar kernel over [red][out] planes."""
    from pulse_amd._lib import EPI_RELU_GRAD, GEMM_OUT_CONTIG
    dev = "cuda:0"
    print(f"{'case':40s} {'x3 us':>9s} {'TF/s':>7s} {'x3p us':>9s} {'TF/s':>7s}")
    for name, rows, m, n, S in [("dW L1 2048x934 over 16384 (split 8)", 16384, 2048, 934, 8), ("dW L2 512x1024 over 16384 x2 (split 8)", 16384, 512, 1024, 8)]:
        batch = 2 if "x2" in name else 1
        dz = torch.randn(rows, batch * m, device=dev)
        x = torch.relu(torch.randn(rows, batch * ((n + 3) // 4 * 4), device=dev))
        n4 = (n + 3) // 4 * 4
        P = (batch * m * n4 + 1023) // 1024 * 1024
        slabs = torch.empty(S, P, device=dev)
        flops = 2.0 * m * n * rows * batch
        f0 = lambda: K.gemm(dz, x, slabs, M=m, N=n, K=rows, lda=batch * m, ldb=batch * n4, ldc=n4, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, batch=batch,
                            stride_a=m, stride_b=n4, stride_c=m * n4, split_k=S, split_stride=P, f32_mode="x3")
        t0 = timed(f0, reps)
        pz, px = K.split_planes(dz), K.split_planes(x)
        f1 = lambda: K.gemm_x3p(pz, px, M=m, N=n, K=rows, C=slabs, ldc=n4, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, batch=batch, stride_a=m,
                                stride_b=((n + 31) // 32 * 32) if batch > 1 else 0, stride_c=m * n4, split_k=S, split_stride=P)
        if batch > 1:                                             # per-net column blocks of X start at multiples of the padded width
            x2 = torch.zeros(rows, batch * ((n + 31) // 32 * 32), device=dev)
            for b in range(batch):
                x2[:, b * ((n + 31) // 32 * 32):b * ((n + 31) // 32 * 32) + n] = x[:, b * n4:b * n4 + n]
            px = K.split_planes(x2)
        t1 = timed(f1, reps)
        print(f"{name:40s} {t0:9.1f} {flops / t0 * 1e-6:7.1f} {t1:9.1f} {flops / t1 * 1e-6:7.1f}", flush=True)
    for name, m, n, k in [("dX L2 16384x1024x512 x2", 16384, 1024, 512)]:
        batch = 2
        dz = torch.randn(m, batch * k, device=dev)
        w = torch.randn(batch * k, n, device=dev) * 0.03
        h = torch.relu(torch.randn(m, batch * n, device=dev))
        c = torch.empty(m, batch * n, device=dev)
        flops = 2.0 * m * n * k * batch
        f0 = lambda: K.gemm(dz, w, c, M=m, N=n, K=k, lda=batch * k, ldb=n, ldc=batch * n, b_layout=GEMM_OUT_CONTIG, batch=batch, stride_a=k, stride_b=k * n, stride_c=n,
                            epilogue=EPI_RELU_GRAD, aux=h, ldaux=batch * n, stride_aux=n, f32_mode="x3")
        t0 = timed(f0, reps)
        pz, pw, ph = K.split_planes(dz), K.split_planes(w), K.split_planes(h)
        cp = K.alloc_planes(m, batch * n, dev)
        f1 = lambda: K.gemm_x3p(pz, pw, M=m, N=n, K=k, Cp=cp, b_layout=GEMM_OUT_CONTIG, batch=batch, stride_a=k, stride_b=k * pw.stride(1), stride_cp=n,
                                epilogue=EPI_RELU_GRAD, aux=ph[0], ldaux=ph.stride(1), stride_aux=n)
        t1 = timed(f1, reps)
        # forward form over W^T planes
        pwt = torch.stack([K.split_planes(w[b * k:(b + 1) * k], transpose=True) for b in range(batch)], 1).reshape(3, batch * n, -1).contiguous()
        f2 = lambda: K.gemm_x3p(pz, pwt, M=m, N=n, K=k, Cp=cp, batch=batch, stride_a=k, stride_b=n * pwt.stride(1), stride_cp=n,
                                epilogue=EPI_RELU_GRAD, aux=ph[0], ldaux=ph.stride(1), stride_aux=n)
        t2 = timed(f2, reps)
        print(f"{name:40s} {t0:9.1f} {flops / t0 * 1e-6:7.1f} {t1:9.1f} {flops / t1 * 1e-6:7.1f}   forward form over W^T planes: {t2:9.1f} us {flops / t2 * 1e-6:7.1f} TF/s (all writing the output's planes)", flush=True)


if __name__ == "__main__":
    if "--backward" in sys.argv:
        sys.argv.remove("--backward")
        backward_forms(30)
        sys.exit(0)

    main()
