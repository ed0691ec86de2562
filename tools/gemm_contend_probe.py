"""Every GEMM tiling of the product path beside another stream's GEMMs: each launch form is run alone once (reference) and then ITERS times with a second
stream issuing x3 and bf16-storage GEMMs; any element that differs from the reference is counted.   python tools/gemm_contend_probe.py   (round 6)"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
from pulse_amd._lib import ACT_RELU, ACT_SILU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
K.F32_MODE = "x3"
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
keep = []


def case(name, **kw):
    d = K.make_gemm_desc(**kw)
    keep.append(kw)
    return (name, d, kw["C"])


cases = []
# wide tile: forward + ReLU, input gradient with aux, weight gradient on 8 slabs with row sums
x, w, b = rnd(4096, 960), rnd(2048, 960) / 30, rnd(2048)
h = torch.empty(4096, 2048, device=dev)
cases.append(case("x3 256x256 forward relu", A=x, B=w, C=h, M=4096, N=2048, K=934, lda=960, ldb=960, ldc=2048, bias=b, activation=ACT_RELU))
K.launch_gemm(*cases[-1][1])
dz, w2, dx = rnd(4096, 1024), rnd(1024, 2048) / 30, torch.empty(4096, 2048, device=dev)
cases.append(case("x3 256x256 input gradient (aux)", A=dz, B=w2, C=dx, M=4096, N=2048, K=1024, lda=1024, ldb=2048, ldc=2048, b_layout=GEMM_OUT_CONTIG,
                  epilogue=EPI_RELU_GRAD, aux=h, ldaux=2048))
slabs = torch.zeros(8, 2048 * 960 + 2048, device=dev)
dzw = rnd(16384, 2048)
xw = rnd(16384, 960)
cases.append(case("x3 256x256 weight gradient, 8 slabs + row sums", A=dzw, B=xw, C=slabs, M=2048, N=960, K=16384, lda=2048, ldb=960, ldc=960, a_layout=GEMM_OUT_CONTIG,
                  b_layout=GEMM_OUT_CONTIG, split_k=8, split_stride=slabs.stride(0), rowsum=slabs, rowsum_off=2048 * 960))
# 128 x 128 / 64 x 128 tiles: rollout-sized forward, batched layer 2, SiLU with pre-activation
xs, hs = rnd(1024, 960), torch.empty(1024, 2048, device=dev)
cases.append(case("x3 128x128 forward relu", A=xs, B=w, C=hs, M=1024, N=2048, K=934, lda=960, ldb=960, ldc=2048, bias=b, activation=ACT_RELU))
K.launch_gemm(*cases[-1][1])
w3, h2 = rnd(1024, 1024) / 30, torch.empty(1024, 1024, device=dev)
cases.append(case("x3 64x128 batched layer", A=hs, B=w3, C=h2, M=1024, N=512, K=1024, lda=2048, ldb=1024, ldc=1024, activation=ACT_RELU, batch=2, stride_a=1024,
                  stride_b=512 * 1024, stride_c=512))
# skinny-N heads forward (one round of 256 workgroups)
hh, wh, heads = rnd(16384, 1024), rnd(2 * 69, 512) / 22, torch.zeros(16384, 144, device=dev)
cases.append(case("x3 skinny-N heads forward", A=hh, B=wh, C=heads, M=16384, N=69, K=512, lda=1024, ldb=512, ldc=144, batch=2, stride_a=512, stride_b=69 * 512, stride_c=72))
# positive control: the opt-in fp32 ReLU bit mask -- forward that WRITES it, input gradient that READS it (the form DESIGN 6 found unsafe)
mk = K.alloc_relu_mask(1024, 1024, dev)
h2m = torch.empty(1024, 1024, device=dev)
cases.append(case("x3 64x128 batched layer writing the bit mask", A=hs, B=w3, C=h2m, M=1024, N=512, K=1024, lda=2048, ldb=1024, ldc=1024, activation=ACT_RELU, batch=2,
                  stride_a=1024, stride_b=512 * 1024, stride_c=512, relu_mask=mk, ld_mask=mk.stride(0), stride_mask=128))
K.launch_gemm(*cases[-1][1])
dyh, whd, dxm = rnd(1024, 144), rnd(2 * 69, 512), torch.empty(1024, 1024, device=dev)
cases.append(case("x3 64x128 input gradient READING the bit mask", A=dyh, B=whd, C=dxm, M=1024, N=512, K=69, lda=144, ldb=512, ldc=1024, b_layout=GEMM_OUT_CONTIG, batch=2,
                  stride_a=72, stride_b=69 * 512, stride_c=512, epilogue=EPI_RELU_GRAD, relu_mask=mk, ld_mask=mk.stride(0), stride_mask=128))
dxa = torch.empty(1024, 1024, device=dev)
cases.append(case("   the same launch reading the activations (aux)", A=dyh, B=whd, C=dxa, M=1024, N=512, K=69, lda=144, ldb=512, ldc=1024, b_layout=GEMM_OUT_CONTIG, batch=2,
                  stride_a=72, stride_b=69 * 512, stride_c=512, epilogue=EPI_RELU_GRAD, aux=h2m, ldaux=1024, stride_aux=512))
# the same pair on the 256 x 256 tile (cfg2's layer-2 input gradient: 16384 x 1024 x 512, both nets; here 4096 rows)
mkw = K.alloc_relu_mask(4096, 2048, dev)
hw_ = torch.empty(4096, 2048, device=dev)
cases.append(case("x3 256x256 forward writing the bit mask", A=x, B=w, C=hw_, M=4096, N=2048, K=934, lda=960, ldb=960, ldc=2048, bias=b, activation=ACT_RELU,
                  relu_mask=mkw, ld_mask=mkw.stride(0)))
K.launch_gemm(*cases[-1][1])
dz2, w2b, dxw = rnd(4096, 1024), rnd(2 * 512, 1024) / 30, torch.empty(4096, 2048, device=dev)
cases.append(case("x3 256x256 input gradient READING the bit mask", A=dz2, B=w2b, C=dxw, M=4096, N=1024, K=512, lda=1024, ldb=1024, ldc=2048, b_layout=GEMM_OUT_CONTIG,
                  batch=2, stride_a=512, stride_b=512 * 1024, stride_c=1024, epilogue=EPI_RELU_GRAD, relu_mask=mkw, ld_mask=mkw.stride(0), stride_mask=256))
# ragged tiles (the epilogue's per-element path): 1000 x 500 per net
mkr = K.alloc_relu_mask(1000, 1024, dev)
hr_ = torch.empty(1000, 1024, device=dev)
xr = rnd(1000, 2048)
cases.append(case("x3 ragged batched layer writing the bit mask", A=xr, B=w3, C=hr_, M=1000, N=500, K=1024, lda=2048, ldb=1024, ldc=1024, activation=ACT_RELU, batch=2,
                  stride_a=1024, stride_b=512 * 1024, stride_c=512, relu_mask=mkr, ld_mask=mkr.stride(0), stride_mask=128))
K.launch_gemm(*cases[-1][1])
dyr, dxr = rnd(1000, 144), torch.empty(1000, 1024, device=dev)
cases.append(case("x3 ragged input gradient READING the bit mask", A=dyr, B=whd, C=dxr, M=1000, N=500, K=69, lda=144, ldb=512, ldc=1024, b_layout=GEMM_OUT_CONTIG, batch=2,
                  stride_a=72, stride_b=69 * 512, stride_c=512, epilogue=EPI_RELU_GRAD, relu_mask=mkr, ld_mask=mkr.stride(0), stride_mask=128))
# ... and with the 256 x 256 tile forced (gemm option 4 = 2): the kernel that serves these launches at 16384 rows
dxw2 = torch.empty(4096, 2048, device=dev)
cases.append(case("[wide forced] input gradient READING the bit mask", A=dz2, B=w2b, C=dxw2, M=4096, N=1024, K=512, lda=1024, ldb=1024, ldc=2048, b_layout=GEMM_OUT_CONTIG,
                  batch=2, stride_a=512, stride_b=512 * 1024, stride_c=1024, epilogue=EPI_RELU_GRAD, relu_mask=mkw, ld_mask=mkw.stride(0), stride_mask=256))
hw2, mkw2 = torch.empty(4096, 2048, device=dev), K.alloc_relu_mask(4096, 2048, dev)
cases.append(case("[wide forced] forward writing the bit mask", A=x, B=w, C=hw2, M=4096, N=2048, K=934, lda=960, ldb=960, ldc=2048, bias=b, activation=ACT_RELU,
                  relu_mask=mkw2, ld_mask=mkw2.stride(0)))
WIDE = {len(cases) - 2, len(cases) - 1}


def launch(i, d):
    if i in WIDE:
        K.gemm_set_option(4, 2)
    K.launch_gemm(*d)
    if i in WIDE:
        K.gemm_set_option(4, 0)


for i, (_, d, _) in enumerate(cases):
    launch(i, d)
torch.cuda.synchronize()
refs = [c[2].clone() for c in cases]
side = torch.cuda.Stream()
a2, b2, c2 = rnd(256, 1960), rnd(512, 1960), torch.empty(256, 512, device=dev)
d_side = K.make_gemm_desc(a2, b2, c2, M=256, N=512, K=1960, lda=1960, ldb=1960, ldc=512, activation=ACT_RELU)
to16 = lambda t: (t.contiguous().view(torch.int32) + 0x8000 >> 16).to(torch.int16)
xs16, ws16, hs16 = to16(rnd(4096, 512)), to16(rnd(1024, 512) / 22), torch.zeros(4096, 1024, dtype=torch.int16, device=dev)
d_side16 = K.make_gemm_x3p_desc(xs16, ws16, planes=1, M=4096, N=1024, K=512, Cp=hs16, activation=ACT_RELU)
iters = int(os.environ.get("ITERS", "400"))
bad = torch.zeros(len(cases), dtype=torch.int64, device=dev)
for it in range(iters):
    with torch.cuda.stream(side):
        for _ in range(4):
            K.launch_gemm(*d_side)
            K.launch_gemm_x3p(*d_side16)
    for i, (name, d, out) in enumerate(cases):
        launch(i, d)
        bad[i] += (out != refs[i]).sum()
torch.cuda.synchronize()
for (name, _, _), n in zip(cases, bad.tolist()):
    print(f"{name:50s} {iters} launches beside a second stream: {n} elements differ from the launch alone")
