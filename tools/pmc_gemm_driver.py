"""Driver for tools/pmc_gemm.sh: a few launches of every GEMM arithmetic on the cfg2 layer-1 forward shape and on 4096^3 (ReLU-like A)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pulse_amd import kernels as K  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
if os.environ.get("PMC_DRIVER") == "b16":
    # the bf16-storage GEMM (pulse_gemm_x3p, planes = 1) on the cfg5 shapes: forward form, weight-gradient form ([red][out] x [red][out], split-K)
    from pulse_amd._lib import ACT_RELU, GEMM_OUT_CONTIG
    for m, n, k in ((16384, 2048, 934), (12288, 1024, 1960)):
        x = K.to_b16(torch.relu(torch.randn(m, k, device=dev)))
        w = K.to_b16(torch.randn(n, k, device=dev) * 0.03)
        cp = K.alloc_b16(m, n, dev)
        bias = torch.randn(n, device=dev)
        for _ in range(6):
            K.gemm_x3p(x, w, M=m, N=n, K=k, Cp=cp, bias=bias, activation=ACT_RELU, planes=1)
    for rows, m, n, S in ((16384, 2048, 960, 4), (16384, 1024, 1984, 4)):
        dz = K.to_b16(torch.randn(rows, m, device=dev))
        x = K.to_b16(torch.relu(torch.randn(rows, n, device=dev)))
        slabs = torch.empty(S, m * n, device=dev)
        for _ in range(6):
            K.gemm_x3p(dz, x, M=m, N=n, K=rows, C=slabs, ldc=n, planes=1, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=S, split_stride=m * n)
    torch.cuda.synchronize()
    sys.exit(0)
for m, n, k in ((16384, 2048, 960), (4096, 4096, 4096)):
    x = torch.relu(torch.randn(m, k, device=dev))
    w = torch.randn(n, k, device=dev) * 0.03
    c = torch.empty(m, n, device=dev)
    px, pw = K.split_planes(x), K.split_planes(w)
    for _ in range(6):
        K.gemm(x, w, c, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, f32_mode="x3")
    for _ in range(6):
        K.gemm_x3p(px, pw, M=m, N=n, K=k, C=c, ldc=n)
    for _ in range(6):
        K.gemm(x, w, c, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, f32_mode="mfma32")
    for _ in range(6):
        K.gemm(x, w, c, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, compute_bf16=True)
torch.cuda.synchronize()
