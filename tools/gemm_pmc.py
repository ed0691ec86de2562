"""Tiny driver for PMC collection: a few launches of our GEMM and of torch.mm at 4096^3 and the cfg2 layer-1 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
big = torch.randn(4096, 4096, device=dev); c = torch.empty(4096, 4096, device=dev)
x = torch.randn(16384, 960, device=dev); w1 = torch.randn(2048, 960, device=dev) / 31; h1 = torch.empty(16384, 2048, device=dev)
for _ in range(6):
    K.gemm(big, big, c, M=4096, N=4096, K=4096, lda=4096, ldb=4096, ldc=4096)
    torch.mm(big, big.t())
    K.gemm(x, w1, h1, M=16384, N=2048, K=960, lda=960, ldb=960, ldc=2048)
    torch.mm(x, w1.t())
torch.cuda.synchronize()
