import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs
agent, rollout = configs.make_agent("cfg2")
for _ in range(2):
    agent.train_epoch()
torch.cuda.synchronize()
# rollout: CPU issue time vs wall
t0 = time.perf_counter(); bd = agent.play_steps(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"play_steps: cpu issue {1e3*(t1-t0):.1f} ms, wall {1e3*(t2-t0):.1f} ms")
agent.set_train(); bd.pop("played_frames")
t0 = time.perf_counter(); agent.prepare_dataset(bd); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"prepare_dataset: cpu {1e3*(t1-t0):.2f} ms wall {1e3*(t2-t0):.2f} ms")
t0 = time.perf_counter()
for _ in range(agent.mini_epochs_num):
    for i in range(len(agent.dataset)):
        agent.train_actor_critic(agent.dataset[i])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"48 minibatch steps: cpu issue {1e3*(t1-t0):.1f} ms ({1e3*(t1-t0)/48:.2f} ms/step), wall {1e3*(t2-t0):.1f} ms ({1e3*(t2-t0)/48:.2f} ms/step)")
# one step with per-phase sync timing
import pulse_amd.kernels as K
d = agent.dataset[0]
def timed(name, fn):
    torch.cuda.synchronize(); a = time.perf_counter(); fn(); torch.cuda.synchronize(); print(f"   {name}: {1e6*(time.perf_counter()-a):.0f} us")
idx, obs_store, act_store, mu_store, old_nlp, adv, old_val, ret = agent._gather_inputs(d)
mb = idx.numel(); net = agent.model; ws = net.workspace(mb, True)
timed("rms gather+norm+update", lambda: agent.running_mean_std.forward(obs_store, row_idx=idx, out=ws["x"], out_cols=net.in_pitch))
timed("forward", lambda: net.forward(ws, mb))
timed("backward", lambda: net.backward(ws, mb))
timed("sqnorm+adam", lambda: (K.sqnorm_partial(net.grad, net.n_flat, agent._sq_partials), K.adam_step(net.flat, net.grad, agent.exp_avg, agent.exp_avg_sq, net.n_flat, lr=1e-5, step=5, max_norm=50.0, sqnorm_partials=agent._sq_partials)))
for i, op in enumerate(ws["plan_fwd"].ops + ws["plan_bwd"].ops):
    if op[0] == 0:
        dd = op[1]
        torch.cuda.synchronize(); a = time.perf_counter(); K.launch_gemm(dd, op[2], op[3]); torch.cuda.synchronize(); t = time.perf_counter() - a
        print(f"   gemm {op[3]:3s} M={dd.M:6d} N={dd.N:5d} K={dd.K:6d} batch={dd.batch} split={dd.split_k:3d}: {1e6*t:7.0f} us  {op[2]/t/1e12:6.1f} TF/s")
