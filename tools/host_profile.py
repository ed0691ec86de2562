"""Dev tool: where does the HOST spend its time in train_epoch?  (cProfile over a few epochs + per-minibatch issue timestamps.)"""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs
gc.collect(); gc.freeze(); gc.disable()
agent, _ = configs.make_agent("cfg2", device="cuda:0", seed=1234, reference="motion_lib")
agent.init_tensors(); agent.obs = agent.env_reset(); agent._tensors_ready = True
for _ in range(3):
    agent.train_epoch()
# host issue time of each minibatch (no sync inside)
orig = agent.train_actor_critic
stamps = []
def wrapped(d):
    t0 = time.perf_counter(); r = orig(d); stamps.append(time.perf_counter() - t0); return r
agent.train_actor_critic = wrapped
for e in range(4):
    stamps.clear()
    info = agent.train_epoch()
    s = sorted(stamps)
    print(f"epoch {e}: update {1e3*info['update_time']:.1f} ms; host issue per minibatch: sum {1e3*sum(stamps):.1f} ms, median {1e3*s[len(s)//2]:.2f}, max {1e3*s[-1]:.2f}, top5 {[round(1e3*x,2) for x in s[-5:]]}", flush=True)
agent.train_actor_critic = orig
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    agent.train_epoch()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
