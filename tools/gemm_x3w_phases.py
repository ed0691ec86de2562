"""Where a launch of the 256 x 256 x3 GEMM (gemm_x3w_kernel) spends its time: per-workgroup shader-cycle and wall-clock (100 MHz) stamps at the
start, after the main loop, after the epilogue has issued its stores and after they are acknowledged.
    python tools/gemm_x3w_phases.py"""
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
    K.F32_MODE = "x3"
    K.gemm_set_option(4, 2)
    for form, m, n, ks in (("fwd", 16384, 2048, (64, 934, 3096)), ("fwd", 16384, 1024, (64, 512, 1024)), ("dw", 2048, 960, (2048,))):
        for k in ks:
            if form == "fwd":
                kp = (k + 31) // 32 * 32
                x = torch.relu(torch.randn(m, kp, device=dev))
                w = torch.randn(n, kp, device=dev) * 0.03
                out = torch.empty(m, n, device=dev)
                d = K.make_gemm_desc(x, w, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=n, bias=torch.randn(n, device=dev), activation=ACT_RELU)
                wgs, nkt = (m // 256) * (n // 256), (k + 15) // 16
            else:
                S = 8
                rows = k * S
                dz, x = torch.randn(rows, m, device=dev), torch.relu(torch.randn(rows, n, device=dev))
                slabs = torch.empty(S, m * n, device=dev)
                d = K.make_gemm_desc(dz, x, slabs, M=m, N=n, K=rows, lda=m, ldb=n, ldc=n, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=S,
                                     split_stride=m * n)
                wgs, nkt = (m // 256) * ((n + 255) // 256) * S, (k + 15) // 16
            for _ in range(10):
                K.launch_gemm(*d)
            torch.cuda.synchronize()
            bufs = [torch.zeros(wgs, 8, dtype=torch.int64, device=dev) for _ in range(4)]
            for b in bufs:
                _lib.check(lib.pulse_gemm_set_debug_buffer(b.data_ptr()), "dbg")
                K.launch_gemm(*d)
            _lib.check(lib.pulse_gemm_set_debug_buffer(None), "dbg")
            torch.cuda.synchronize()
            b = bufs[2].cpu().double()
            nxt = bufs[3].cpu().double()
            w0, w1, w2, w3 = b[:, 1] / 100, b[:, 3] / 100, b[:, 5] / 100, b[:, 7] / 100          # us
            cyc = b[:, 2] - b[:, 0]
            ghz = float((cyc / (w1 - w0)).median()) * 1e-3
            ideal = nkt * 96 * 32.0
            med = lambda v: float(v.median())
            print(f"{form} {m}x{n} K={k:5d} ({wgs} wgs, {nkt} k-tiles): span {float(w3.max() - w0.min()):6.1f} us | start spread {float(w0.max() - w0.min()):5.1f} | "
                  f"prologue + main loop {med(w1 - w0):6.1f} us = {med(cyc):9.0f} cyc at {ghz:.2f} GHz, MFMA pipe {ideal / med(cyc):.3f} busy | "
                  f"epilogue issue {med(w2 - w1):5.1f} | store drain {med(w3 - w2):5.1f} | "
                  f"per-WG total {med(w3 - w0):6.1f} (max {float((w3 - w0).max()):6.1f}) | gap to next launch {float(nxt[:, 1].min() / 100 - w3.max()):5.1f}", flush=True)
    K.gemm_set_option(4, 0)


if __name__ == "__main__":
    main()
