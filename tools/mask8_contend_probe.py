"""The bf16-storage twin of tools/mask_contend_probe.py: does a relu-grad launch of pulse_gemm_x3p that reads the sign BYTES give the bits of the launch
that re-reads the bf16 activations when another stream's GEMMs run beside it?    python tools/mask8_contend_probe.py   (SIDE=0: no second stream)"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
to16 = lambda t: (t.contiguous().view(torch.int32) + 0x8000 >> 16).to(torch.int16)
rnd = lambda *s: torch.randn(*s, generator=g)
res = []
for (m, n, k, kg) in ((256, 512, 512, 96), (4096, 1024, 512, 512)):
    x16, w16 = to16(rnd(m, k)).to(dev), to16(rnd(n, k) / math.sqrt(k)).to(dev)
    h16 = torch.zeros(m, n, dtype=torch.int16, device=dev)
    mask = K.alloc_relu_mask8(m, n, dev)
    K.gemm_x3p(x16, w16, planes=1, M=m, N=n, K=k, Cp=h16, activation=ACT_RELU, relu_mask8=mask)
    dy16, w2 = to16(rnd(m, kg)).to(dev), to16(rnd(kg, n)).to(dev)
    o_mask, o_aux = torch.zeros(m, n, dtype=torch.int16, device=dev), torch.zeros(m, n, dtype=torch.int16, device=dev)
    cs1, cs2 = [torch.zeros(K.gemm_x3p_row_tiles(m, n, 1), n, device=dev) for _ in range(2)]
    kw = dict(planes=1, M=m, N=n, K=kg, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD)
    d_mask = K.make_gemm_x3p_desc(dy16, w2, Cp=o_mask, relu_mask8=mask, out_colsum=cs1, **kw)
    d_aux = K.make_gemm_x3p_desc(dy16, w2, Cp=o_aux, aux=h16, ldaux=n, out_colsum=cs2, **kw)
    h16b, maskb = h16.clone(), mask.clone()                     # (descriptors hold raw pointers: the buffers must outlive them)
    d_fwd = K.make_gemm_x3p_desc(x16, w16, planes=1, M=m, N=n, K=k, Cp=h16b, activation=ACT_RELU, relu_mask8=maskb)
    K.launch_gemm_x3p(*d_aux)
    torch.cuda.synchronize()
    ref, ref_cs = o_aux.clone(), cs2.clone()
    side = torch.cuda.Stream()
    a2, b2, c2 = torch.randn(256, 1960, device=dev), torch.randn(512, 1960, device=dev), torch.empty(256, 512, device=dev)
    d_side = K.make_gemm_desc(a2, b2, c2, M=256, N=512, K=1960, lda=1960, ldb=1960, ldc=512, activation=ACT_RELU)
    xs, ws_ = to16(rnd(4096, 512)).to(dev), to16(rnd(1024, 512) / 22).to(dev)
    hs = torch.zeros(4096, 1024, dtype=torch.int16, device=dev)
    d_side16 = K.make_gemm_x3p_desc(xs, ws_, planes=1, M=4096, N=1024, K=512, Cp=hs, activation=ACT_RELU)
    bad = torch.zeros(4, dtype=torch.int64, device=dev)
    with_side = os.environ.get("SIDE", "1") == "1"
    iters = int(os.environ.get("ITERS", "2000"))
    for it in range(iters):
        if with_side:
            with torch.cuda.stream(side):
                K.launch_gemm(*d_side)
                K.launch_gemm_x3p(*d_side16)
                K.launch_gemm(*d_side)
        o_mask.fill_(0x7fc0)
        o_aux.fill_(0x7fc0)
        K.launch_gemm_x3p(*d_mask)
        K.launch_gemm_x3p(*d_aux)
        K.launch_gemm_x3p(*d_fwd)                       # the forward's round-once rows as well: mask bytes and activations must stay what they were
        bad[0] += (o_mask != ref).sum()
        bad[1] += (o_aux != ref).sum()
        bad[2] += (cs1 != ref_cs).sum() + (cs2 != ref_cs).sum()
        bad[3] += (h16b != h16).sum() + (maskb != mask).sum()
    torch.cuda.synchronize()
    print(f"{m} x {n} x {kg}: {'side stream' if with_side else 'alone'}, {iters} iterations: sign-byte variant mismatches {int(bad[0])}, aux variant {int(bad[1])}, column sums {int(bad[2])}, forward outputs / sign bytes {int(bad[3])}")
