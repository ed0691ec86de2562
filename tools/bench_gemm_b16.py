"""Time the bf16-storage GEMM (pulse_gemm_x3p, planes = 1) against the fp32-storage bf16 kernel (pulse_gemm_f32, COMPUTE_BF16) on the
cfg5 training shapes (run on the GPU box):    python tools/bench_gemm_b16.py [--reps 30]
Forward, input-gradient and weight-gradient forms of the actor / critic MLP (minibatch 16384, batched pair) and the discriminator
(3 x 4096 forward rows, 4 x 4096 stacked gradient rows, 1960-wide AMP window)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulse_amd import kernels as K  # noqa: E402
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG  # noqa: E402


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


def r4(v):
    return (v + 3) // 4 * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--tile", type=int, default=0, help="gemm option 3: 0 automatic, 1 never the 256 x 256 tile, 2 always")
    a = ap.parse_args()
    K.gemm_set_option(3, a.tile)
    dev = "cuda:0"
    S = 8
    print(f"{'case':40s} {'f32-storage us':>15s} {'TF/s':>7s} {'bf16-storage us':>16s} {'TF/s':>7s} {'ratio':>6s}")
    rows = []
    # ---- forward: relu(X W^T + b)
    for name, m, n, k in [("fwd a2c L1 16384x1024x934", 16384, 1024, 934), ("fwd a2c L2 16384x512x1024", 16384, 512, 1024), ("fwd disc L1 12288x1024x1960", 12288, 1024, 1960),
                          ("fwd disc L2 12288x512x1024", 12288, 512, 1024), ("fwd rollout 8192x1024x934", 8192, 1024, 934)]:
        x = torch.relu(torch.randn(m, r4(k), device=dev)); x[:, k:] = 0
        w = torch.randn(n, r4(k), device=dev) * 0.03; w[:, k:] = 0
        bias = torch.randn(n, device=dev)
        c = torch.empty(m, r4(n), device=dev)
        f0 = lambda: K.gemm(x, w, c, M=m, N=n, K=k, lda=r4(k), ldb=r4(k), ldc=r4(n), bias=bias, activation=ACT_RELU, compute_bf16=True)
        px, pw, cp = K.to_b16(x[:, :k].contiguous()), K.to_b16(w[:, :k].contiguous()), K.alloc_b16(m, n, dev)
        f1 = lambda: K.gemm_x3p(px, pw, M=m, N=n, K=k, Cp=cp, bias=bias, activation=ACT_RELU, planes=1)
        rows.append((name, 2.0 * m * n * k, timed(f0, a.reps), timed(f1, a.reps)))
    # ---- input gradient: (dZ W) * relu'(H)
    for name, m, n, k in [("dX a2c L2 16384x1024x512", 16384, 1024, 512), ("dX disc L2 12288x1024x512", 12288, 1024, 512), ("dX disc pen 4096x1960x1024", 4096, 1960, 1024)]:
        dz = torch.randn(m, k, device=dev)
        w = torch.randn(k, r4(n), device=dev) * 0.03
        h = torch.relu(torch.randn(m, r4(n), device=dev))
        c = torch.empty(m, r4(n), device=dev)
        f0 = lambda: K.gemm(dz, w, c, M=m, N=n, K=k, lda=k, ldb=r4(n), ldc=r4(n), b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=h, ldaux=r4(n), compute_bf16=True)
        pz, pw, ph, cp = K.to_b16(dz), K.to_b16(w[:, :n].contiguous()), K.to_b16(h[:, :n].contiguous()), K.alloc_b16(m, n, dev)
        f1 = lambda: K.gemm_x3p(pz, pw, M=m, N=n, K=k, Cp=cp, planes=1, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=ph, ldaux=ph.stride(0))
        rows.append((name, 2.0 * m * n * k, timed(f0, a.reps), timed(f1, a.reps)))
    # ---- weight gradient: dZ^T X over the batch, split-K slabs
    for name, rws, m, n in [("dW a2c L1 1024x934 over 16384", 16384, 1024, 934), ("dW a2c L2 512x1024 over 16384", 16384, 512, 1024), ("dW disc L1 1024x1960 over 16384", 16384, 1024, 1960),
                            ("dW disc L2 512x1024 over 16384", 16384, 512, 1024)]:
        dz = torch.randn(rws, m, device=dev)
        x = torch.relu(torch.randn(rws, r4(n), device=dev))
        pstride = (m * r4(n) + 1023) // 1024 * 1024
        slabs = torch.empty(S, pstride, device=dev)
        f0 = lambda: K.gemm(dz, x, slabs, M=m, N=n, K=rws, lda=m, ldb=r4(n), ldc=r4(n), a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=S, split_stride=pstride,
                            compute_bf16=True)
        pz, px = K.to_b16(dz), K.to_b16(x[:, :n].contiguous())
        f1 = lambda: K.gemm_x3p(pz, px, M=m, N=n, K=rws, C=slabs, ldc=r4(n), planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=S, split_stride=pstride)
        rows.append((name, 2.0 * m * n * rws, timed(f0, a.reps), timed(f1, a.reps)))
    for name, flops, t0, t1 in rows:
        print(f"{name:40s} {t0:15.1f} {flops / t0 * 1e-6:7.1f} {t1:16.1f} {flops / t1 * 1e-6:7.1f} {t0 / t1:6.2f}", flush=True)


if __name__ == "__main__":
    main()
