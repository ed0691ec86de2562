"""Per-kernel roofline of the HBM / latency-bound kernels at cfg2 sizes (dev tool; the official bench is bench.py).
Each kernel is timed in isolation with HIP events on the launch stream; algorithmic bytes per launch are the ones DESIGN.md
section 3 states.  Prints a markdown table (achieved GB/s vs the 8 TB/s HBM3E spec peak)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pulse_amd import configs, kernels as K, ops
from pulse_amd._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS

dev = "cuda:0"
PEAK = 8.0e12


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


rows = []


def report(name, nbytes, t, unit_desc):
    rows.append(f"| `{name}` | {unit_desc} | {nbytes / 1e6:.2f} | {t * 1e6:.1f} | {nbytes / t / 1e9:.0f} | {nbytes / t / PEAK:.3f} |")


agent, _ = configs.make_agent("cfg2", device=dev, seed=1, reference="motion_lib")
agent.init_tensors()
agent.obs = agent.env_reset()
task = agent.vec_env.env.task
n, T = task.num_envs, agent.horizon_length
for _ in range(3):
    task.step(torch.zeros(n, 69, device=dev))
full = PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS
t = timeit(lambda: task._im_step(full, inc=0))
report("im_step_kernel (library mode, reward+reset+obs)", n * (1248 + 552 + 4 * 1920 + 3840 + 40 + 1800 + 40), t, f"{n} envs")
lib = task._motion_lib
ids = task._sampled_motion_ids
times = torch.rand(n, device=dev) * task._motion_len_env
out = {}
t = timeit(lambda: lib.query(ids, times, task._global_offset, out=out))
report("motion_state_kernel (get_motion_state)", n * (2 * 1908 + 1848 + 20), t, f"{n} queries")
rb, dp, dv = task.sim.rigid_body_state, task.sim.dof_pos, task.sim.dof_vel
n2 = 8192
rb2, dp2, dv2 = rb.repeat(2, 1, 1), dp.repeat(2, 1), dv.repeat(2, 1)
key = torch.tensor([7, 3, 22, 17], dtype=torch.int32, device=dev)
amp_out = torch.zeros(n2, 232, device=dev)
t = timeit(lambda: ops.build_amp_observations_smpl(rb2, dp2, dv2, key, out=amp_out))
report("amp_obs_kernel", n2 * (52 + 552 + 48 + 928), t, f"{n2} envs")
mb = agent.minibatch_size
eb = agent.experience_buffer
idx = torch.randperm(n * T, device=dev)[:mb]
ws = agent.model.workspace(mb, train=True)
rms = agent.running_mean_std
t = timeit(lambda: rms.forward(eb.flat("obses"), row_idx=idx, out=ws["x"], out_cols=agent.model.in_pitch, update=False))
report("rms_normalize_vec4_kernel (gather + normalise, no moments)", mb * ((934 + 960) * 4 + 8), t, f"{mb} rows")
t = timeit(lambda: rms.forward(eb.flat("obses"), row_idx=idx, out=ws["x"], out_cols=agent.model.in_pitch, update=True))
report("rms_normalize_vec4 + rms_update (with fp64 moments)", mb * ((934 + 960) * 4 + 8), t, f"{mb} rows")
td = eb.tensor_dict
t = timeit(lambda: ops.discount_values(td["dones"], td["values"], td["rewards"], td["next_values"], 0.99, 0.95, return_returns=True))
report("gae_kernel", n * T * 21, t, f"{T}x{n} elements")
net = agent.model
t = timeit(lambda: [K.reduce_slabs(net._slabs, ns, net.n_flat, cnt, net.grad, scale=1.0, slabs_off=off, out_off=off) for off, cnt, ns in net._slab_regions(ws)])
report("reduce_slabs_kernel (8 slabs)", net.n_flat * 4 * (net.split_k + 1), t, f"{net.n_flat} params")
sq, gn = torch.zeros(256, device=dev), torch.zeros(1, device=dev)


def opt():
    K.sqnorm_partial(net.grad, net.n_flat, sq)
    K.adam_step(net.flat, net.grad, agent.exp_avg, agent.exp_avg_sq, net.n_flat, lr=0.0, step=1, max_norm=50.0, sqnorm_partials=sq, grad_norm_out=gn)


t = timeit(opt)
report("sqnorm_partial + adam_kernel", net.n_flat * 4 * (1 + 4 + 3), t, f"{net.n_flat} params")
# the PPO loss head on one minibatch (gathered actions / old mu, mu | value of the heads, d mu | d value out)
td0 = eb.tensor_dict
lp = torch.zeros(1024, 8, device=dev)
wsl = net.workspace(mb, train=True)
wsl["heads"].normal_()
fl = lambda k: eb.flat(k)
adv_flat = torch.randn(n * T, device=dev)


def loss():
    K.ppo_loss(mu=wsl["mu"], mu_stride=wsl["mu"].stride(0), value=wsl["val"], value_stride=wsl["val"].stride(0), logstd=net.sigma, old_logstd=net.sigma,
               idx=idx, actions=fl("actions"), actions_stride=fl("actions").stride(0), old_mu=fl("mus"), old_mu_stride=fl("mus").stride(0),
               old_neglogp=fl("neglogpacs").reshape(-1), advantages=adv_flat, old_values=fl("values").reshape(-1), returns=adv_flat, rows=mb, num_actions=69,
               e_clip=0.2, critic_coef=5.0, bounds_loss_coef=10.0, clip_value=False, dmu=wsl["dmu"], dmu_stride=wsl["dmu"].stride(0),
               dvalue=wsl["dval"], dvalue_stride=wsl["dval"].stride(0), partials=lp)


t = timeit(loss)
# (the Python wrapper fills a 200-byte argument struct per call: back-to-back launches are host-bound at ~14 us; rocprofv3 --kernel-trace of this
# tool gives the kernel: 9.6 us hoisted, 10.6 us plain at 1024 workgroups; 10.4 / 14.0 at 512 -- profiles/r06_ab_runs.txt)
report("ppo_loss_kernel (losses + d mu | d value; host-bound here)", mb * (69 * 4 * 4 + 8 + 6 * 4), t, f"{mb} rows")
K.gemm_set_option(8, 1)
t = timeit(loss)
K.gemm_set_option(8, 0)
report("ppo_loss_kernel, plain per-sample form (gemm option 8)", mb * (69 * 4 * 4 + 8 + 6 * 4), t, f"{mb} rows")
vm = agent.value_mean_std
wsr = net.workspace(n, train=False)
mask = torch.zeros(n, dtype=torch.bool, device=dev)


def rec():
    K.rollout_record(rewards=task.rew_buf, dones=task.reset_buf, terminate=task._terminate_buf, value_raw=wsr["val"], value_stride=wsr["val"].stride(0),
                     value_mean=vm.running_mean, value_var=vm.running_var, value_eps=vm.epsilon, buf_rewards=eb.phys["rewards"][:, 0],
                     buf_next_values=eb.phys["next_values"][:, 0], buf_dones=eb.phys["dones"][:, 0], env_stride=T, current_rewards=agent.current_rewards,
                     current_lengths=agent.current_lengths, meter_rewards=agent.game_rewards.state, meter_lengths=agent.game_lengths.state,
                     meter_max_size=agent.games_to_track, done_mask=mask)


t = timeit(rec)
report("rollout_record_kernel (one workgroup)", n * 46, t, f"{n} envs")
t = timeit(task.sim.simulate_and_refresh)
report("kinematic_sim_kernel (physics stand-in)", n * (24 * 13 * 12 + 69 * 4 * 8), t, f"{n} envs")
# ---- the fused env step over recorded reference frames at 8 x cfg2's width: time per env when every CU has 16 workgroups to run
# (it scales linearly from 4096 envs: the kernel is bound by its own instruction stream -- ~12 us per wave of two envs --, not by HBM)
from pulse_amd import synthetic as syn  # noqa: E402
for n_big in (32768,):                                           # (at 4096 envs this Python wrapper, not the kernel, sets the pace of back-to-back launches)
    d = syn.env_step_inputs(syn.make_generator(5), n_big)
    to = lambda x: x.to(dev)
    rbb = to(d["rb"])
    rn, rx = {k: to(v) for k, v in d["ref_now"].items()}, {k: to(v) for k, v in d["ref_next"].items()}
    kw = dict(what=full, ref_now=rn, ref_next=rx, dof_force=to(d["dof_force"]), dof_vel=to(d["dof_vel"]), progress=to(d["progress"]), pass_time=to(d["pass_time"]),
              track_ids=list(range(24)), reset_ids=syn.RESET_BODY_IDS, term_dist=torch.full((24,), 0.25, device=dev),
              obs=torch.zeros(n_big, 960, device=dev), obs_cols=960, rew=torch.zeros(n_big, device=dev), rew_raw=torch.zeros(n_big, 5, device=dev),
              reset=torch.zeros(n_big, dtype=torch.int64, device=dev), terminate=torch.zeros(n_big, dtype=torch.int64, device=dev))
    t = timeit(lambda: ops.im_step(rbb, **kw))
    report("im_step_kernel (recorded reference frames, reward+reset+obs)", n_big * 8080, t, f"{n_big} envs (SURVEY 8d: 8 080 B per env-step)")
# ---- round 3: terrain / trajectory step (height-map gather), PULSE VAE head kernels, downstream-task step
from pulse_amd._lib import TASK_OBS, TASK_REWARD, TASK_RESET  # noqa: E402
try:
    agent_t, _ = configs.make_agent("terrain_z", device=dev, seed=1, num_envs_override=4096)
    agent_t.init_tensors()
    agent_t.obs = agent_t.env_reset()
    tt = agent_t.vec_env.env.task
    nt = tt.num_envs
    nh = tt.get_task_obs_size_detail()["heightmap"]
    t = timeit(lambda: tt._task_step(TASK_OBS | TASK_REWARD | TASK_RESET))
    # per env: bodies 1 248 + dof force / vel 552 + trajectory vertices 101 x 12 + three int16 cells per height point, task obs (20 + nh) x 4, reward 12, flags 16, progress 8
    report("traj_step_kernel (trajectory samples + height map + reward + reset)", nt * (1248 + 552 + 101 * 12 + nh * 6 + (20 + nh) * 4 + 12 + 16 + 8), t, f"{nt} envs, {nh} height points")
except Exception as e:                                           # keep the older rows if a config cannot be built
    print("terrain_z skipped:", e)
E, A, S_ = 32, 69, 358
rows_v = 16384
heads, pheads = torch.randn(rows_v, 2 * E, device=dev), torch.randn(rows_v, 2 * E, device=dev)
eps, xin = torch.randn(rows_v, E, device=dev), torch.randn(rows_v, 960, device=dev)
ain, cin = torch.zeros(rows_v, 392, device=dev), torch.zeros(rows_v, 392, device=dev)
t = timeit(lambda: K.vae_embed(heads, xin, ain, rows=rows_v, embedding_size=E, self_obs_size=S_, z_col=360, eps=eps, cin=cin, clamp=True, clamp_max=2.0))
report("vae_embed_kernel (re-parameterise + decoder / critic input rows)", rows_v * (2 * E * 4 + E * 4 + S_ * 4 + (S_ + E) * 4 + S_ * 4), t, f"{rows_v} rows")
pred, gt, dmu = torch.randn(rows_v, 72, device=dev), torch.randn(rows_v, 72, device=dev), torch.zeros(rows_v, 72, device=dev)
prog = (torch.arange(rows_v, device=dev) % 32).to(torch.int64)
part = torch.zeros(256, 8, device=dev)
t = timeit(lambda: K.vae_kin_loss(pred, gt, heads, pheads, prog, dmu, part, rows=rows_v, num_actions=A, embedding_size=E, horizon=32, clamp=True, clamp_max=2.0,
                                  use_ar1=True, use_regu=True))
report("vae_kin_loss_kernel (RMSE + KL vs learned prior + AR(1) + regulariser)", rows_v * (2 * A * 4 + 4 * E * 4 + 8 + A * 4), t, f"{rows_v} rows")
dz, dzh, dph = torch.randn(rows_v, E, device=dev), torch.zeros(rows_v, 2 * E, device=dev), torch.zeros(rows_v, 2 * E, device=dev)
t = timeit(lambda: K.vae_head_backward(heads, dzh, rows=rows_v, embedding_size=E, horizon=32, pheads=pheads, dpheads=dph, eps=eps, dz=dz, progress=prog, clamp=True,
                                       clamp_max=2.0, c_kl=1e-4, c_ar1=1e-5, c_regu=1e-7))
report("vae_head_backward_kernel", rows_v * (4 * E * 4 + 2 * E * 4 + 8 + 4 * E * 4), t, f"{rows_v} rows")
print("| kernel | units per launch | algorithmic MB per launch | us per launch | GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|---|")
print("\n".join(rows))
