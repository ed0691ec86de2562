"""Micro-benchmark of pulse_gemm_f32 on the config-2 shapes (dev tool, not the official bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import kernels as K
from pulse_amd._lib import GEMM_OUT_CONTIG, GEMM_RED_CONTIG, ACT_RELU, EPI_RELU_GRAD

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, flops, t, t_ref=None):
    extra = f"   torch {flops / t_ref / 1e12:7.1f} TF/s ({t_ref * 1e6:8.1f} us)" if t_ref else ""
    print(f"{name:34s} {t * 1e6:9.1f} us  {flops / t / 1e12:7.1f} TF/s{extra}", flush=True)


M = int(os.environ.get("M", 16384))
x = torch.randn(M, 960, device=dev)
w1 = torch.randn(2048, 960, device=dev) / 31
b1 = torch.randn(2048, device=dev)
h1 = torch.empty(M, 2048, device=dev)
t = timeit(lambda: K.gemm(x, w1, h1, M=M, N=2048, K=960, lda=960, ldb=960, ldc=2048, bias=b1, activation=ACT_RELU))
tr = timeit(lambda: torch.relu(torch.addmm(b1, x, w1.t())))
report("fwd L1 16384x2048x960 (+bias,relu)", 2 * M * 2048 * 960, t, tr)

w2 = torch.randn(2, 512, 1024, device=dev) / 32
b2 = torch.randn(2, 512, device=dev)
h2 = torch.empty(M, 1024, device=dev)
t = timeit(lambda: K.gemm(h1, w2, h2, M=M, N=512, K=1024, lda=2048, ldb=1024, ldc=1024, bias=b2, activation=ACT_RELU, batch=2,
                          stride_a=1024, stride_b=512 * 1024, stride_c=512, stride_bias=512))
h1a = h1[:, :1024].contiguous()
tr = timeit(lambda: torch.relu(torch.addmm(b2[0], h1a, w2[0].t()))) * 2
report("fwd L2 2x(16384x512x1024)", 2 * 2 * M * 512 * 1024, t, tr)

dz2 = torch.randn(M, 1024, device=dev)
dz1 = torch.empty(M, 2048, device=dev)
t = timeit(lambda: K.gemm(dz2, w2, dz1, M=M, N=1024, K=512, lda=1024, ldb=1024, ldc=2048, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD,
                          aux=h1, ldaux=2048, batch=2, stride_a=512, stride_b=512 * 1024, stride_c=1024, stride_aux=1024))
dz2a = dz2[:, :512].contiguous()
tr = timeit(lambda: torch.mm(dz2a, w2[0])) * 2
report("dX L2 2x(16384x1024x512)", 2 * 2 * M * 512 * 1024, t, tr)

for S in (4, 8, 16):
    slab = 2048 * 960
    slabs = torch.empty(S, slab, device=dev)
    t = timeit(lambda: K.gemm(dz1, x, slabs, M=2048, N=960, K=M, lda=2048, ldb=960, ldc=960, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                              split_k=S, split_stride=slab))
    out = torch.empty(slab, device=dev)
    t2 = timeit(lambda: K.reduce_slabs(slabs, S, slab, slab, out))
    tr = timeit(lambda: torch.mm(dz1.t(), x))
    report(f"dW1 2048x960x16384 split{S}", 2 * M * 2048 * 960, t, tr)
    print(f"     reduce_slabs S={S}: {t2 * 1e6:.1f} us", flush=True)

S = 8
slab = 2 * 512 * 1024
slabs = torch.empty(S, slab, device=dev)
t = timeit(lambda: K.gemm(dz2, h1, slabs, M=512, N=1024, K=M, lda=1024, ldb=2048, ldc=1024, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                          batch=2, stride_a=512, stride_b=1024, stride_c=512 * 1024, split_k=S, split_stride=slab))
tr = timeit(lambda: torch.mm(dz2a.t(), h1a)) * 2
report("dW2 2x(512x1024x16384) split8", 2 * 2 * M * 512 * 1024, t, tr)

# rollout-size forward
Mr = 4096
xr = x[:Mr]
t = timeit(lambda: K.gemm(xr, w1, h1, M=Mr, N=2048, K=960, lda=960, ldb=960, ldc=2048, bias=b1, activation=ACT_RELU))
report("rollout fwd L1 4096x2048x960", 2 * Mr * 2048 * 960, t)
big = torch.randn(4096, 4096, device=dev)
c = torch.empty(4096, 4096, device=dev)
t = timeit(lambda: K.gemm(big, big, c, M=4096, N=4096, K=4096, lda=4096, ldb=4096, ldc=4096))
tr = timeit(lambda: torch.mm(big, big.t()))
report("4096^3", 2 * 4096 ** 3, t, tr)
