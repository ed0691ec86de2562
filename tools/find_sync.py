import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulse_amd import configs, kernels as K
agent, rollout = configs.make_agent("cfg2")
for _ in range(2):
    agent.train_epoch()
bd = agent.play_steps(); agent.set_train(); bd.pop("played_frames"); agent.prepare_dataset(bd)
torch.cuda.synchronize()
acc = {}
def T(name, fn):
    a = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0) + time.perf_counter() - a; return r
net = agent.model
for rep in range(12):
    d = T("dataset[i]", lambda: agent.dataset[rep % 8])
    idx, obs_store, act_store, mu_store, old_nlp, adv, old_val, ret = T("gather_inputs", lambda: agent._gather_inputs(d))
    mb = idx.numel(); ws = net.workspace(mb, True); ap = net.a_pitch
    T("rms", lambda: agent.running_mean_std.forward(obs_store, row_idx=idx, out=ws["x"], out_cols=net.in_pitch))
    T("forward", lambda: net.forward(ws, mb))
    T("ppo_loss", lambda: K.ppo_loss(mu=ws["heads"], mu_stride=2 * ap, value=ws["val"], value_stride=2 * ap, logstd=net.sigma, old_logstd=net.sigma, idx=idx,
                   actions=act_store, actions_stride=act_store.stride(0), old_mu=mu_store, old_mu_stride=mu_store.stride(0),
                   old_neglogp=old_nlp, advantages=adv, old_values=old_val, returns=ret, rows=mb, num_actions=69,
                   e_clip=0.2, critic_coef=5, bounds_loss_coef=10, clip_value=False,
                   dmu=ws["dheads"], dmu_stride=2 * ap, dvalue=ws["dheads"][:, ap:], dvalue_stride=2 * ap, partials=agent._loss_partials))
    for i, op in enumerate(ws["plan_bwd"].ops):
        if op[0] == 0:
            T(f"bwd{i}:gemm {op[3]} {op[1].M}x{op[1].N}x{op[1].K}", lambda: K.launch_gemm(op[1], op[2], op[3]))
        else:
            T(f"bwd{i}:{op[3]}", lambda: K._lib.check(op[1](*op[2], K._stream()), op[3]))
    T("final reduce", lambda: [K.reduce_slabs(net._slabs, ns, net.n_flat, cnt, net.grad, scale=1.0, slabs_off=off, out_off=off) for off, cnt, ns in net._slab_regions(ws)])
    T("sqnorm", lambda: K.sqnorm_partial(net.grad, net.n_flat, agent._sq_partials))
    T("adam", lambda: K.adam_step(net.flat, net.grad, agent.exp_avg, agent.exp_avg_sq, net.n_flat, lr=1e-5, step=5 + rep, max_norm=50.0, sqnorm_partials=agent._sq_partials, grad_norm_out=agent._grad_norm))
    T("info torch ops", lambda: (agent._loss_partials.sum(0) / mb, agent._grad_norm.clone()))
torch.cuda.synchronize()
for k, v in acc.items():
    print(f"{k:50s} {1e6 * v / 12:9.1f} us/step")
print("total", 1e6 * sum(acc.values()) / 12)
