"""Dev tool: per-epoch update time vs the summed GEMM time of that epoch (HIP events), to tell GPU-speed changes from host stalls."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs, kernels
gc.collect(); gc.freeze(); gc.disable()
agent, _ = configs.make_agent("cfg2", device="cuda:0", seed=1234, reference="motion_lib")
agent.init_tensors(); agent.obs = agent.env_reset(); agent._tensors_ready = True
for _ in range(2):
    agent.train_epoch()
prof = kernels.PROFILER
for e in range(8):
    prof.start()
    info = agent.train_epoch()
    prof.stop()
    s = prof.summary()
    t = sum(v[1] for v in s.values()); f = sum(v[2] for v in s.values())
    print(f"epoch {e}: play {1e3*info['play_time']:.1f} ms update {1e3*info['update_time']:.1f} ms | GEMM {1e3*t:.1f} ms {f/t/1e12:.1f} TF/s | "
          + " ".join(f"{k}:{v[2]/v[1]/1e12:.0f}" for k, v in s.items()), flush=True)
