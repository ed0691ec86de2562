"""Does a HIP graph shorten the gap between dependent kernels?  A chain of 24 small dependent launches (a GEMM feeding the next, the shape of the
cfg2 layer-2 pair, and a chain of 5 us elementwise kernels) timed eagerly and as one captured graph replay.
    python tools/graph_gap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pulse_amd import kernels as K  # noqa: E402
from pulse_amd._lib import ACT_RELU  # noqa: E402

dev = torch.device("cuda:0")
K.F32_MODE = "x3"
M = 16384
h = [torch.relu(torch.randn(M, 1024, device=dev)) for _ in range(2)]
w = torch.randn(2, 512, 1024, device=dev) * 0.03
w2 = torch.randn(2, 1024 // 2, 512, device=dev) * 0.03
descs = []
for i in range(24):
    # 16384 x 512 x 1024 (batch 2) reading h[i % 2] -> writing the first 1024 columns' worth into h[(i + 1) % 2] (dependent chain)
    descs.append(K.make_gemm_desc(h[i % 2], w, h[(i + 1) % 2], M=M, N=512, K=1024, lda=1024, ldb=1024, ldc=1024, activation=ACT_RELU, batch=2, stride_a=0,
                                  stride_b=512 * 1024, stride_c=512))
small = [torch.randn(1 << 20, device=dev) for _ in range(2)]


def gemm_chain():
    for d in descs:
        K.launch_gemm(*d)


def small_chain():
    for i in range(24):
        torch.add(small[i % 2], 1.0, out=small[(i + 1) % 2])


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, fn in (("24 dependent x3 GEMMs (2 x 16384 x 512 x 1024)", gemm_chain), ("24 dependent 4 MB elementwise adds", small_chain)):
    eager = timeit(fn)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    graph = timeit(g.replay)
    print(f"{name}: eager {eager:8.1f} us per chain ({eager / 24:6.2f} per launch), graph replay {graph:8.1f} us ({graph / 24:6.2f} per launch), "
          f"{(eager - graph) / 24:5.2f} us saved per launch", flush=True)
