"""Every distinct fp32 (x3) GEMM launch of one training epoch of a bench config, timed on both tilings (gemm option 4: 1 narrow, 2 wide, 0 the
launcher's choice) with the epoch's own descriptors and buffers.
    python tools/gemm_shapes_ab.py cfg3 [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pulse_amd import configs, kernels as K  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
agent, _ = configs.make_agent(cfg, device="cuda:0", reference="motion_lib")
agent.train_epoch()
seen, order = {}, []
orig = K.launch_gemm


def rec(d, flops=0.0, tag="fwd", stream=None):
    key = (d.M, d.N, d.K, d.batch, d.split_k, d.a_layout, d.b_layout, d.epilogue, d.activation, d.compute_type)
    if key not in seen:
        seen[key] = [d, flops, tag, 0]
        order.append(key)
    seen[key][3] += 1
    return orig(d, flops, tag, stream)


K.launch_gemm = rec
agent.train_epoch()
K.launch_gemm = orig
torch.cuda.synchronize()


def timeit(d):
    for _ in range(2):
        orig(d)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        orig(d)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


rows = []
for key in order:
    d, flops, tag, n = seen[key]
    if key[9] != 2:
        continue
    t = {}
    for opt in (1, 2, 0):
        K.gemm_set_option(4, opt)
        t[opt] = timeit(d)
    K.gemm_set_option(4, 0)
    rows.append((n * t[0], key, tag, n, t, flops))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
best = sum(r[3] * min(r[4][1], r[4][2]) for r in rows)
print(f"# {cfg}: {len(rows)} distinct x3 launches, {tot / 1e3:.1f} ms per epoch with the launcher's choice, {best / 1e3:.1f} ms with the better tiling everywhere")
print("# M N K batch split layouts epi act | launches | narrow us | wide us | auto us | TF/s auto | share")
for share, key, tag, n, t, flops in rows:
    lay = "KK" if (key[5], key[6]) == (0, 0) else "KM" if (key[5], key[6]) == (0, 1) else "MM"
    pick = "wide" if abs(t[0] - t[2]) < abs(t[0] - t[1]) else "narrow"
    flag = "" if min(t[1], t[2]) > 0.97 * t[0] else "   <-- other tiling better"
    print(f"{key[0]:6d} {key[1]:6d} {key[2]:6d} b{key[3]} s{key[4]:<2d} {lay} e{key[7]} a{key[8]} | {n:4d} | {t[1]:8.1f} | {t[2]:8.1f} | {t[0]:8.1f} ({pick}) | "
          f"{flops / t[0] / 1e6:6.1f} | {share / tot:.3f}{flag}")
