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
    print(f"epoch {e}: play {info['play_time']*1e3:.1f} ms update {info['update_time']*1e3:.1f} ms total {info['total_time']*1e3:.1f} ms "
          f"-> {n / info['total_time']:.0f} env-steps/s | a_loss {torch.stack(info['actor_loss']).mean().item():.5f} "
          f"c_loss {torch.stack(info['critic_loss']).mean().item():.5f} b_loss {torch.stack(info['b_loss']).mean().item():.5f} "
          f"kl {torch.stack(info['kl']).mean().item():.6f} gnorm {torch.stack(info['grad_norm']).mean().item():.4f}", flush=True)
print("reward mean", agent.experience_buffer.tensor_dict['rewards'].mean().item(), "dones", agent.experience_buffer.tensor_dict['dones'].float().mean().item())
