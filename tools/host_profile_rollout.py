"""Dev tool: HOST time of the rollout loop (play_steps) -- is the rollout bound by launch issue or by the GPU?  Times play_steps with and
without a device sync before reading the clock, and cProfiles it."""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs
gc.collect(); gc.freeze(); gc.disable()
agent, _ = configs.make_agent("cfg2", device="cuda:0", seed=1234, reference="motion_lib")
agent.init_tensors(); agent.obs = agent.env_reset(); agent._tensors_ready = True
for _ in range(3):
    agent.train_epoch()
for e in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.play_steps()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"play_steps: host returns after {1e3 * (t1 - t0):.2f} ms, device done after {1e3 * (t2 - t0):.2f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    agent.play_steps()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
