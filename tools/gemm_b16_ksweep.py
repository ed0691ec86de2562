"""Reduction-length sweep of the bf16-storage GEMM (pulse_gemm_x3p, planes = 1) with descriptors built ONCE (no host work between launches):
time = fixed part (launch, prologue, epilogue) + stages x per-stage cost.  python tools/gemm_b16_ksweep.py [--reps 50]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulse_amd import kernels as K  # noqa: E402
from pulse_amd._lib import ACT_RELU, GEMM_OUT_CONTIG  # noqa: E402


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
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = "cuda:0"
    for form in ("fwd", "dw"):
        for m, n in ((16384, 1024), (16384, 2048)) if form == "fwd" else ((1024, 2048), (2048, 2048)):
            for tile in (1, 2):
                K.gemm_set_option(3, tile)
                line = []
                for k in (64, 256, 512, 1024, 2048, 4096, 8192):
                    if form == "fwd":
                        x, w = K.to_b16(torch.relu(torch.randn(m, k, device=dev))), K.to_b16(torch.randn(n, k, device=dev) * 0.03)
                        cp = K.alloc_b16(m, n, dev)
                        bias = torch.randn(n, device=dev)
                        d, fl, tag = K.make_gemm_x3p_desc(x, w, M=m, N=n, K=k, Cp=cp, bias=bias, activation=ACT_RELU, planes=1)
                        flops = 2.0 * m * n * k
                    else:
                        S = 8
                        rows = k * S
                        dz, x = K.to_b16(torch.randn(rows, m, device=dev)), K.to_b16(torch.relu(torch.randn(rows, n, device=dev)))
                        slabs = torch.empty(S, m * n, device=dev)
                        d, fl, tag = K.make_gemm_x3p_desc(dz, x, M=m, N=n, K=rows, C=slabs, ldc=n, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                                                          split_k=S, split_stride=m * n)
                        flops = 2.0 * m * n * rows
                    t = timed(lambda: K.launch_gemm_x3p(d, fl, tag), a.reps)
                    line.append(f"K={k}: {t:6.1f} us {flops / t * 1e-6:6.0f} TF/s")
                print(f"{form} {m}x{n} tile={'256x128' if tile == 1 else '256x256'} (K per split for dw) | " + " | ".join(line), flush=True)
    K.gemm_set_option(3, 0)


if __name__ == "__main__":
    main()
