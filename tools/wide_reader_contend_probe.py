"""cfg2-scale twin of gemm_contend_probe.py for the 256 x 256 kernel: the layer-2 and heads input gradients at 16384 rows, reading the ReLU bit mask / the
activations, beside a second stream's GEMMs (round 6, DESIGN 6: 0 differing elements in 3 000 launches each, also beside a GEMM-hammering process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
K.F32_MODE = "x3"
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
m = 16384
cases = []
# layer-1 forward writing the mask (both nets, N = 2048), then the layer-2 input gradient reading it (16384 x 1024 x 512 per net)
x, w, b = rnd(m, 960), rnd(2048, 960) / 30, rnd(2048)
h, mk = torch.empty(m, 2048, device=dev), K.alloc_relu_mask(m, 2048, dev)
d_f = K.make_gemm_desc(x, w, h, M=m, N=2048, K=934, lda=960, ldb=960, ldc=2048, bias=b, activation=ACT_RELU, relu_mask=mk, ld_mask=mk.stride(0))
K.launch_gemm(*d_f)
dz, w2 = rnd(m, 1024), rnd(1024, 1024) / 30
o1, o2 = torch.empty(m, 2048, device=dev), torch.empty(m, 2048, device=dev)
kw = dict(M=m, N=1024, K=512, lda=1024, ldb=1024, ldc=2048, b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=512, stride_b=512 * 1024, stride_c=1024, epilogue=EPI_RELU_GRAD)
cases.append(("layer-2 input gradient reading the bit mask", K.make_gemm_desc(dz, w2, o1, relu_mask=mk, ld_mask=mk.stride(0), stride_mask=256, **kw), o1))
cases.append(("   ... reading the activations", K.make_gemm_desc(dz, w2, o2, aux=h, ldaux=2048, stride_aux=1024, **kw), o2))
h2, mk2 = torch.empty(m, 1024, device=dev), K.alloc_relu_mask(m, 1024, dev)
xh = rnd(m, 2048)
w3 = rnd(1024, 1024) / 30
d_f2 = K.make_gemm_desc(xh, w3, h2, M=m, N=512, K=1024, lda=2048, ldb=1024, ldc=1024, activation=ACT_RELU, batch=2, stride_a=1024, stride_b=512 * 1024, stride_c=512,
                        relu_mask=mk2, ld_mask=mk2.stride(0), stride_mask=128)
K.launch_gemm(*d_f2)
dy, wh = rnd(m, 144), rnd(2 * 69, 512)
o3, o4 = torch.empty(m, 1024, device=dev), torch.empty(m, 1024, device=dev)
kw = dict(M=m, N=512, K=69, lda=144, ldb=512, ldc=1024, b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=72, stride_b=69 * 512, stride_c=512, epilogue=EPI_RELU_GRAD)
cases.append(("heads input gradient reading the bit mask", K.make_gemm_desc(dy, wh, o3, relu_mask=mk2, ld_mask=mk2.stride(0), stride_mask=128, **kw), o3))
cases.append(("   ... reading the activations", K.make_gemm_desc(dy, wh, o4, aux=h2, ldaux=1024, stride_aux=512, **kw), o4))
tiles = []
for _, d, _ in cases:
    K.launch_gemm(*d)
    tiles.append(K._lib.load().pulse_gemm_last_tile())
torch.cuda.synchronize()
refs = [c[2].clone() for c in cases]
assert torch.equal(refs[0], refs[1]) and torch.equal(refs[2], refs[3])
side = torch.cuda.Stream()
a2, b2, c2 = rnd(4096, 960), rnd(2048, 960), torch.empty(4096, 2048, device=dev)
d_side = K.make_gemm_desc(a2, b2, c2, M=4096, N=2048, K=934, lda=960, ldb=960, ldc=2048, activation=ACT_RELU)
iters = int(os.environ.get("ITERS", "600"))
bad = torch.zeros(len(cases), dtype=torch.int64, device=dev)
for it in range(iters):
    with torch.cuda.stream(side):
        for _ in range(3):
            K.launch_gemm(*d_side)
    for i, (_, d, out) in enumerate(cases):
        K.launch_gemm(*d)
        bad[i] += (out != refs[i]).sum()
torch.cuda.synchronize()
for (name, _, _), n, t in zip(cases, bad.tolist(), tiles):
    print(f"{name:48s} tile code {t}: {iters} launches beside a second stream: {n} elements differ")
