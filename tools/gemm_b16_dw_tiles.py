"""Weight-gradient launches of cfg5 on the two bf16-storage tilings at equal workgroup count: 256 x 128 tiles with S slabs vs 256 x 256 tiles with 2 S
slabs (descriptors built once).  python tools/gemm_b16_dw_tiles.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pulse_amd import kernels as K  # noqa: E402
from pulse_amd._lib import GEMM_OUT_CONTIG  # noqa: E402
from gemm_b16_ksweep import timed  # noqa: E402


def main():
    dev = "cuda:0"
    for name, rows, m, n in (("a2c L1 (both nets stacked) 2048 x 934 over 16384", 16384, 2048, 934), ("disc L1 1024 x 1960 over 16384", 16384, 1024, 1960),
                             ("disc L1 1024 x 1960 over 12288", 12288, 1024, 1960)):
        dz, x = K.to_b16(torch.randn(rows, m, device=dev)), K.to_b16(torch.relu(torch.randn(rows, n, device=dev)))
        ldc = (n + 3) // 4 * 4
        out = []
        for tile, S in ((1, 4), (1, 8), (2, 8), (2, 4)):
            K.gemm_set_option(3, tile)
            slabs = torch.empty(S, m * ldc, device=dev)
            d, fl, tag = K.make_gemm_x3p_desc(dz, x, M=m, N=n, K=rows, C=slabs, ldc=ldc, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                                              split_k=S, split_stride=m * ldc)
            t = timed(lambda: K.launch_gemm_x3p(d, fl, tag), 50)
            red = torch.empty(m * ldc, device=dev)
            tr = timed(lambda: K.reduce_slabs(slabs, S, m * ldc, m * ldc, red), 50)
            out.append(f"{'256x128' if tile == 1 else '256x256'} S={S}: {t:6.1f} us ({2.0 * m * n * rows / t * 1e-6:5.0f} TF/s) + reduce {tr:5.1f}")
        print(f"{name:50s} | " + " | ".join(out), flush=True)
    K.gemm_set_option(3, 0)


if __name__ == "__main__":
    main()
