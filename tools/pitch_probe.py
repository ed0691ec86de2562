"""Does the row pitch of the layer-1 operands matter?  The 256 x 256 x3 forward 16384 x 2048 x 934 timed over operand pitches (floats); the obs /
weight pitch of the product path is 960.   python tools/pitch_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulse_amd import kernels as K  # noqa: E402
from pulse_amd._lib import ACT_RELU, GEMM_OUT_CONTIG  # noqa: E402

K.F32_MODE = "x3"
dev = "cuda:0"


def timeit(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


m, n, k = 16384, 2048, 934
b = torch.randn(n, device=dev)
out = torch.empty(m, n, device=dev)
for kp in (936, 944, 960, 964, 968, 976, 992, 1008, 1024, 1040, 1056, 1088):
    x = torch.zeros(m, kp, device=dev); x[:, :k] = torch.relu(torch.randn(m, k, device=dev))
    w = torch.zeros(n, kp, device=dev); w[:, :k] = torch.randn(n, k, device=dev) * 0.03
    t = timeit(lambda: K.gemm(x, w, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=n, bias=b, activation=ACT_RELU))
    # the layer-1 weight gradient reads x as the [red][out] operand: 2048 x 960 outputs over 16384 rows, 8 slabs
    dz = torch.randn(m, n, device=dev)
    slabs = torch.empty(8, n * kp, device=dev)
    t2 = timeit(lambda: K.gemm(dz, x, slabs, M=n, N=kp, K=m, lda=n, ldb=kp, ldc=kp, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=8, split_stride=n * kp))
    print(f"pitch {kp:5d} floats ({kp * 4:5d} B): forward {t:7.1f} us = {2.0 * m * n * k / t / 1e6:6.1f} TFLOP/s | weight gradient {t2:7.1f} us", flush=True)
