"""The RCCL path, executed: a one-rank `nccl` process group on the 1-GPU box (round-4 verdict, missing #5 / next #7).

Every multi-rank test so far ran gloo on a shared device; `parallel.py` defaults to `nccl` (= RCCL on ROCm) and that backend had never run.
This test forces DistContext(enabled=True, backend="nccl") with WORLD_SIZE=1 in a subprocess and drives the data-parallel control flow of the
reference (`setup_algo` broadcast, `optimizer.synchronize()` = bucketed async all-reduce beside the remaining GEMMs, `wait_gradients` before the
optimiser step, `sync_stats` / `average_value` at the epoch's end: phc/learning/common_agent.py:112-127,224-247,465-471) on an AMP agent whose
discriminator chain runs on its side stream -- two streams enqueue around the collectives, which is what the first 8-GPU run would otherwise test
first.  It cannot show scaling.  It shows that the path neither hangs nor mis-orders: the parameters after three optimiser steps match a
non-distributed agent's on the same data (the 1 / world_size scale is 1; the two paths differ only in the association of the gradient-norm sum)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker():
    sys.path.insert(0, ROOT)
    import torch
    from pulse_amd import configs as C
    from pulse_amd.parallel import DistContext

    def run(dist_ctx):
        torch.manual_seed(11)
        kw = dict(device="cuda:0", seed=5, num_envs_override=256, minibatch_size=2048, amp_minibatch_size=2048, horizon_length=16, mini_epochs=1,
                  amp_obs_demo_buffer_size=8192, amp_replay_buffer_size=8192, amp_batch_size=256)
        if dist_ctx is not None:
            kw.update(multi_gpu=True, dist=dist_ctx, overlap_allreduce=True)
        agent, _ = C.make_agent("cfg5", **kw)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        agent._tensors_ready = True
        if dist_ctx is not None:
            agent.dist.setup_algo(agent.model.flat, (agent.model.sigma, agent.exp_avg, agent.exp_avg_sq))
        batch = agent.play_steps()
        batch.pop("played_frames")
        agent.set_train()
        agent.prepare_dataset(batch)
        n = min(3, len(agent.dataset))
        agent._begin_loss_ring(n)
        for i in range(n):
            agent.train_actor_critic(agent.dataset[i])
        agent._end_loss_ring()
        extra = {}
        if dist_ctx is not None:
            frames = agent.dist.sync_stats(agent._stat_modules(), agent.batch_size)
            kl = agent.dist.average_value(torch.tensor([0.25], device="cuda:0"), "ep_kls")
            extra = {"frames": frames, "batch": agent.batch_size, "kl": float(kl.item())}
        torch.cuda.synchronize()
        return agent, extra

    ctx = DistContext(enabled=True, backend="nccl")
    a, extra = run(ctx)
    st = ctx.stats()
    bws = ctx.backend_world_size()
    b, _ = run(None)
    res = {"backend": st["backend"], "backend_world_size": bws, "calls": st["calls"], "bytes": st["bytes"], "side_stream": a._side_stream() is not None,
           "mixed_precision": bool(a.mixed_precision), **extra}
    for name, x, y in (("policy", a.model.flat, b.model.flat), ("disc", a.disc.flat, b.disc.flat)):
        x, y = x.double(), y.double()
        res[name + "_finite"] = bool(torch.isfinite(x).all())
        res[name + "_max_abs_diff"] = float((x - y).abs().max())
        res[name + "_max_abs"] = float(y.abs().max())
        res[name + "_moved"] = float((y - y.mean()).abs().max()) > 0
    ctx.shutdown()
    print("RCCL_RESULT " + json.dumps(res), flush=True)


def test_nccl_group_of_one_runs_the_data_parallel_step():
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29641", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.pop("PULSE_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env,
                       cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_RESULT ")]
    assert p.returncode == 0 and line, p.stdout[-3000:]
    r = json.loads(line[-1][len("RCCL_RESULT "):])
    print(f"[rccl] backend {r['backend']}, backend_world_size {r['backend_world_size']}, {r['calls']} gradient all-reduces ({r['bytes'] / 1e6:.1f} MB), "
          f"side stream {r['side_stream']}, mixed precision {r['mixed_precision']}; max |param diff| vs the non-distributed agent: policy "
          f"{r['policy_max_abs_diff']:.2e}, disc {r['disc_max_abs_diff']:.2e}")
    assert r["backend"] == "nccl" and r["backend_world_size"] == 1
    assert r["calls"] >= 3 and r["bytes"] > 0                                # every optimiser step went through sync_gradients
    assert r["side_stream"] and r["mixed_precision"]                          # the discriminator chain really ran beside the collectives
    assert r["frames"] == r["batch"] and abs(r["kl"] - 0.25) < 1e-7           # sync_stats / average_value over one rank are the identity
    for name in ("policy", "disc"):
        assert r[name + "_finite"]
        # Adam steps of 5e-5-scale learning rates: a mis-ordered (stale or half-reduced) gradient moves parameters by far more than this
        assert r[name + "_max_abs_diff"] <= 2e-6 * max(1.0, r[name + "_max_abs"]), r


if __name__ == "__main__" and "--worker" in sys.argv:
    _worker()
