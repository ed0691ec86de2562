"""Does a relu-grad launch of the fp32 x3 GEMM that takes its derivative from the BIT MASK give the bits of the launch that re-reads the activations
when another stream's GEMMs run beside it?  (round-6 finding: no -- a few elements per thousand launches come out wrong (garbage in one 16-lane quarter of a
wave) with a concurrent stream, 0 alone and 0 for the aux-reading launch; AMPAgent therefore keeps its policy network off the bit masks.)
    python tools/mask_contend_probe.py               # SIDE=0: no second stream; FILL=-1 / 0: constant masks; ITERS=n"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
K.F32_MODE = "x3"
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
m, n, k = 256, 512, 512
x = torch.randn(m, 2 * k, generator=g).to(dev)
w = (torch.randn(2 * n, k, generator=g) / 22).to(dev)
h = torch.empty(m, 2 * n, device=dev)
mask = K.alloc_relu_mask(m, 2 * n, dev)
# forward (batched halves) writing the mask
K.gemm(x, w, h, M=m, N=n, K=k, lda=2 * k, ldb=k, ldc=2 * n, activation=ACT_RELU, batch=2, stride_a=k, stride_b=n * k, stride_c=n,
       relu_mask=mask, ld_mask=mask.stride(0), stride_mask=n // 4)
fillv = os.environ.get("FILL")
if fillv is not None:
    mask.fill_(int(fillv))            # -1: every bit set; 0: none
    h.fill_(1.0 if int(fillv) else -1.0)
dy = torch.randn(m, 2 * 72, generator=g).to(dev)
w2 = torch.randn(2 * 69, n, generator=g).to(dev)              # [red][out]
o_mask, o_aux = torch.empty(m, 2 * n, device=dev), torch.empty(m, 2 * n, device=dev)
kw = dict(M=m, N=n, K=69, lda=2 * 72, ldb=n, ldc=2 * n, b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=72, stride_b=69 * n, stride_c=n, epilogue=EPI_RELU_GRAD)
d_mask = K.make_gemm_desc(dy, w2, o_mask, relu_mask=mask, ld_mask=mask.stride(0), stride_mask=n // 4, **kw)
d_aux = K.make_gemm_desc(dy, w2, o_aux, aux=h, ldaux=2 * n, stride_aux=n, **kw)
side = torch.cuda.Stream()
a2, b2, c2 = torch.randn(256, 1960, device=dev), torch.randn(512, 1960, device=dev), torch.empty(256, 512, device=dev)
d_side = K.make_gemm_desc(a2, b2, c2, M=256, N=512, K=1960, lda=1960, ldb=1960, ldc=512, activation=ACT_RELU)
K.launch_gemm(*d_aux)
torch.cuda.synchronize()
ref = o_aux.clone()
bad = torch.zeros(3, dtype=torch.int64, device=dev)
seen = []
iters = int(os.environ.get("ITERS", "3000"))
with_side = os.environ.get("SIDE", "1") == "1"
for it in range(iters):
    if with_side:
        with torch.cuda.stream(side):
            for _ in range(3):
                K.launch_gemm(*d_side)
    o_mask.fill_(7.0)
    K.launch_gemm(*d_mask)
    o_aux.fill_(9.0)
    K.launch_gemm(*d_aux)
    bad[0] += (o_mask != ref).sum()
    bad[1] += (o_aux != ref).sum()
    bad[2] += ((o_mask != ref) & (o_mask == 0)).sum()
    mm = o_mask != ref
    if len(seen) < 6 and bool(mm.any()):
        idx = torch.nonzero(mm)
        seen.append((it, idx.shape[0], idx[:6].tolist(), o_mask[mm][:6].tolist(), ref[mm][:6].tolist()))
torch.cuda.synchronize()
for r in seen:
    print(r)
print("side stream" if with_side else "alone", "iterations", iters, "mask-variant mismatches", int(bad[0]), "(of which zeros:", int(bad[2]), ") aux-variant mismatches", int(bad[1]))
