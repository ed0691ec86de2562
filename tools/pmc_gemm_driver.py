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
if os.environ.get("PMC_DRIVER") == "x3w":
    # the two tilings of the x3 fp32 GEMM (gemm option 4: 1 = 128 x 128 only, 2 = 256 x 256) on the cfg2 update shapes: layer-1 forward, layer-2
    # forward pair, layer-2 input gradient pair, layer-1 weight gradient (8 slabs), and 4096^3
    from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
    M = 16384
    x = torch.relu(torch.randn(M, 960, device=dev))
    w1 = torch.randn(2048, 960, device=dev) * 0.03
    h1 = torch.empty(M, 2048, device=dev)
    w2 = torch.randn(2, 512, 1024, device=dev) * 0.03
    h2 = torch.empty(M, 1024, device=dev)
    dz2 = torch.randn(M, 1024, device=dev)
    dz1 = torch.empty(M, 2048, device=dev)
    slabs = torch.empty(8, 2048 * 960, device=dev)
    big = torch.relu(torch.randn(4096, 4096, device=dev))
    cb = torch.empty(4096, 4096, device=dev)
    for opt in (1, 2):
        K.gemm_set_option(4, opt)
        for _ in range(5):
            K.gemm(x, w1, h1, M=M, N=2048, K=934, lda=960, ldb=960, ldc=2048, activation=ACT_RELU, f32_mode="x3")
        for _ in range(5):
            K.gemm(h1, w2, h2, M=M, N=512, K=1024, lda=2048, ldb=1024, ldc=1024, activation=ACT_RELU, batch=2, stride_a=1024, stride_b=512 * 1024, stride_c=512,
                   f32_mode="x3")
        for _ in range(5):
            K.gemm(dz2, w2, dz1, M=M, N=1024, K=512, lda=1024, ldb=1024, ldc=2048, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=h1, ldaux=2048, batch=2,
                   stride_a=512, stride_b=512 * 1024, stride_c=1024, stride_aux=1024, f32_mode="x3")
        for _ in range(5):
            K.gemm(dz1, x, slabs, M=2048, N=960, K=M, lda=2048, ldb=960, ldc=960, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=8,
                   split_stride=2048 * 960, f32_mode="x3")
        for _ in range(5):
            K.gemm(big, big, cb, M=4096, N=4096, K=4096, lda=4096, ldb=4096, ldc=4096, f32_mode="x3")
    K.gemm_set_option(4, 0)
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
