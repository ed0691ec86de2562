import time, torch
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
def t(fn, n=20):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); return f"median {1e6*ts[n//2]:.0f} us max {1e6*ts[-1]:.0f} us"
ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
print("Event.synchronize on a completed event:", t(ev.synchronize))
print("Event.query on a completed event      :", t(ev.query))
print("torch.cuda.synchronize, idle device   :", t(torch.cuda.synchronize))
def wait_kernel_sync():
    for _ in range(50): x.add_(1.0)
    e = torch.cuda.Event(); e.record(); e.synchronize()
print("50 small kernels + Event.synchronize  :", t(wait_kernel_sync))
def wait_kernel_spin():
    for _ in range(50): x.add_(1.0)
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
print("50 small kernels + query spin         :", t(wait_kernel_spin))
def wait_dev_sync():
    for _ in range(50): x.add_(1.0)
    torch.cuda.synchronize()
print("50 small kernels + cuda.synchronize   :", t(wait_dev_sync))
import os
print("HIP env:", {k: v for k, v in os.environ.items() if "HIP" in k or "HSA" in k or "AMD" in k})
