import sys, math, torch
sys.path.insert(0, ".")
from pulse_amd import kernels as K
from pulse_amd._lib import GEMM_OUT_CONTIG
K.F32_MODE = "x3"
dev = "cuda:0"
def timeit(f, it=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (m, n, k, batch, lay) in [(16384, 69, 512, 2, "KK"), (16384, 69, 1024, 1, "KK"), (16384, 64, 512, 1, "KK"), (16384, 32, 3096, 1, "KM"), (16384, 64, 160, 1, "KK"), (32768, 69, 512, 2, "KK")]:
    if lay == "KK":
        x = torch.randn(m, batch * k, device=dev); w = torch.randn(batch * n, k, device=dev) / math.sqrt(k); out = torch.zeros(m, batch * 72, device=dev)
        f = lambda: K.gemm(x, w, out, M=m, N=n, K=k, lda=batch * k, ldb=k, ldc=batch * 72, batch=batch, stride_a=k, stride_b=n * k, stride_c=72)
    else:
        x = torch.randn(m, k, device=dev); w = torch.randn(k, 392, device=dev); out = torch.zeros(m, n, device=dev)
        f = lambda: K.gemm(x, w, out, M=m, N=n, K=k, lda=k, ldb=392, ldc=n, b_layout=GEMM_OUT_CONTIG)
    t = {}
    for opt in (1, 0):
        K.gemm_set_option(6, opt); t[opt] = timeit(f)
    K.gemm_set_option(6, 0)
    gb = (m * k * batch * 4) / 1e9
    print(f"{m:6d} x {n:3d} x {k:5d} b{batch} {lay}: old {t[1]:7.1f} us   skinny {t[0]:7.1f} us   ({gb / t[0] * 1e6 / 1e3:.2f} TB/s of A)")
if "--exp" in sys.argv:
    m, n, k, batch = 16384, 69, 512, 2
    x = torch.randn(m, batch * k, device=dev); w = torch.randn(batch * n, k, device=dev) / math.sqrt(k); out = torch.zeros(m, batch * 72, device=dev)
    f = lambda: K.gemm(x, w, out, M=m, N=n, K=k, lda=batch * k, ldb=k, ldc=batch * 72, batch=batch, stride_a=k, stride_b=n * k, stride_c=72)
    names = {0: "full", 1: "no MFMA", 2: "no A split", 4: "no B staging", 8: "no fragment reads", 16: "no A loads", 3: "no MFMA, no A split", 30: "MFMA only", 31: "nothing"}
    for e in (0, 1, 2, 4, 8, 16, 3, 30, 31):
        K.gemm_set_option(7, e)
        print(f"exp {e:2d} ({names[e]}): {timeit(f):7.1f} us")
    K.gemm_set_option(7, 0)
