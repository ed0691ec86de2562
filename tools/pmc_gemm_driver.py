"""Driver for tools/pmc_gemm.sh: a few launches of every GEMM arithmetic on the cfg2 layer-1 forward shape and on 4096^3 (ReLU-like A)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pulse_amd import kernels as K  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
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
