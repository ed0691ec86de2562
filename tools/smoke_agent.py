import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time()
agent, rollout = configs.make_agent(name)
print("setup", time.time() - t0, flush=True)
for e in range(epochs):
    info = agent.train_epoch()
    n = agent.batch_size
    keys = [k for k in ("actor_loss", "critic_loss", "kl", "kin_loss", "kin_action_loss", "kin_KLD", "kin_ar1", "grad_norm") if k in info]
    stats = " ".join(f"{k} {torch.stack([torch.as_tensor(v).float().reshape(-1)[0] for v in info[k]]).mean().item():.5f}" for k in keys)
    print(f"epoch {e}: play {info['play_time']*1e3:.1f} ms update {info['update_time']*1e3:.1f} ms total {info['total_time']*1e3:.1f} ms "
          f"-> {n / info['total_time']:.0f} env-steps/s | {stats}", flush=True)
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
