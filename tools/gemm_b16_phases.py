"""Where a bf16-storage GEMM launch (256 x 256 tile, gemm_b16w_kernel) spends its time: per-workgroup wall-clock stamps (100 MHz) at the
start, when the first stage has landed, after the main loop, after the epilogue has issued its stores and after they are acknowledged.
    python tools/gemm_b16_phases.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulse_amd import _lib, kernels as K  # noqa: E402
from pulse_amd._lib import ACT_RELU, GEMM_OUT_CONTIG  # noqa: E402


def main():
    dev = "cuda:0"
    lib = _lib.load()
    K.gemm_set_option(3, 2)
    for form, m, n in (("fwd", 16384, 1024), ("fwd", 16384, 2048), ("dw", 1024, 2048)):
        for k in (64, 1024, 4096):
            if form == "fwd":
                x, w = K.to_b16(torch.relu(torch.randn(m, k, device=dev))), K.to_b16(torch.randn(n, k, device=dev) * 0.03)
                cp = K.alloc_b16(m, n, dev)
                d, fl, tag = K.make_gemm_x3p_desc(x, w, M=m, N=n, K=k, Cp=cp, bias=torch.randn(n, device=dev), activation=ACT_RELU, planes=1)
                wgs = (m // 256) * (n // 256)
            else:
                S = 8
                rows = k * S
                dz, x = K.to_b16(torch.randn(rows, m, device=dev)), K.to_b16(torch.relu(torch.randn(rows, n, device=dev)))
                slabs = torch.empty(S, m * n, device=dev)
                d, fl, tag = K.make_gemm_x3p_desc(dz, x, M=m, N=n, K=rows, C=slabs, ldc=n, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                                                  split_k=S, split_stride=m * n)
                wgs = (m // 256) * (n // 256) * S
            for _ in range(10):
                K.launch_gemm_x3p(d, fl, tag)
            torch.cuda.synchronize()
            bufs = [torch.zeros(wgs, 8, dtype=torch.int64, device=dev) for _ in range(4)]
            for b in bufs:
                _lib.check(lib.pulse_gemm_set_debug_buffer(b.data_ptr()), "dbg")
                K.launch_gemm_x3p(d, fl, tag)
            _lib.check(lib.pulse_gemm_set_debug_buffer(None), "dbg")
            torch.cuda.synchronize()
            t = [b.cpu().double() / 100.0 for b in bufs]              # us
            b = t[2]
            med = lambda v: float(v.median())
            gap = float(t[3][:, 0].min() - t[2][:, 4].max())
            print(f"{form} {m}x{n} K={k:5d} ({wgs} workgroups): span {float(b[:, 4].max() - b[:, 0].min()):6.1f} us | start spread {float(b[:, 0].max() - b[:, 0].min()):5.1f} | "
                  f"prologue {med(b[:, 1] - b[:, 0]):5.1f} | main loop {med(b[:, 2] - b[:, 1]):6.1f} | epilogue issue {med(b[:, 3] - b[:, 2]):5.1f} | "
                  f"store drain {med(b[:, 4] - b[:, 3]):5.1f} | 2nd half: image written at +{med(b[:, 6] - b[:, 2]):5.1f}, stores issued at +{med(b[:, 7] - b[:, 2]):5.1f} | per-WG total {med(b[:, 4] - b[:, 0]):6.1f} (max {float((b[:, 4] - b[:, 0]).max()):6.1f}) | gap to next launch {gap:5.1f}", flush=True)
    K.gemm_set_option(3, 0)


if __name__ == "__main__":
    main()
