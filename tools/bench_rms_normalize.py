"""Time pulse_rms_normalize (+ pulse_rms_update) at the cfg2 / cfg5 shapes for several workgroup counts (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pulse_amd import kernels as K  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = "cuda:0"
    for rows, cols, total, gather in [(16384, 934, 131072, True), (4096, 934, 4096, False), (12288, 1960, 200000, True), (16384, 934, 262144, True)]:
        pitch = (cols + 31) // 32 * 32
        x = torch.randn(total, pitch, device=dev)
        idx = torch.randperm(total, device=dev)[:rows].contiguous() if gather else None
        y = torch.empty(rows, pitch, device=dev)
        mean = torch.zeros(cols, dtype=torch.float64, device=dev)
        var = torch.ones(cols, dtype=torch.float64, device=dev)
        cnt = torch.ones((), dtype=torch.float64, device=dev)
        line = f"rows {rows:6d} x {cols:4d} gather={gather!s:5s} "
        for nb in (128, 256, 512, 1024, 2048, 4096):
            if nb * 4 > rows:
                continue
            part = torch.zeros(nb, 2, cols, dtype=torch.float64, device=dev)
            t0 = timed(lambda: K.rms_normalize(x, mean, var, rows=rows, cols=cols, x_stride=pitch, y=y, y_stride=pitch, y_cols=pitch, row_idx=idx, num_blocks=nb))
            t1 = timed(lambda: K.rms_normalize(x, mean, var, rows=rows, cols=cols, x_stride=pitch, y=y, y_stride=pitch, y_cols=pitch, row_idx=idx, moment_partials=part))
            t2 = timed(lambda: K.rms_update(mean, var, cnt, part, cols, 1000.0, rows))
            line += f"| nb {nb}: {t0:5.1f} / {t1:5.1f} + {t2:4.1f} us "
        gb = rows * cols * 8 / 1e3
        print(line + f"| {gb / 1e3:.0f} MB moved", flush=True)


if __name__ == "__main__":
    main()
