"""Dev tool: HOST time of the update phase (calc_gradients per minibatch) vs the device's -- is the update bound by launch issue or by the GPU?
    python tools/host_profile_update.py [cfg5]"""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
gc.collect(); gc.freeze(); gc.disable()
agent, _ = configs.make_agent(cfg, device="cuda:0", seed=1234, reference="motion_lib")
agent.init_tensors(); agent.obs = agent.env_reset(); agent._tensors_ready = True
for _ in range(3):
    agent.train_epoch()
orig = agent.calc_gradients
acc = [0.0, 0]


def timed(d):
    t = time.perf_counter()
    orig(d)
    acc[0] += time.perf_counter() - t
    acc[1] += 1


agent.calc_gradients = timed
for e in range(3):
    acc[0], acc[1] = 0.0, 0
    info = agent.train_epoch()
    print(f"{cfg}: update wall {1e3 * info['update_time']:.1f} ms, host inside calc_gradients {1e3 * acc[0]:.1f} ms over {acc[1]} minibatches "
          f"({1e6 * acc[0] / max(1, acc[1]):.0f} us each), play {1e3 * info['play_time']:.1f} ms", flush=True)
agent.calc_gradients = orig
pr = cProfile.Profile()
pr.enable()
agent.train_epoch()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
