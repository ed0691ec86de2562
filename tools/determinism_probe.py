"""Run-to-run determinism probe of the cfg1 pipeline (run on the GPU box):

    python tools/determinism_probe.py --repeats 6                 # one FRESH PROCESS per repeat, same seeds
    python tools/determinism_probe.py --repeats 6 --contend       # the same while a second process hammers the GPU with GEMMs

(Repeats inside one process are not comparable: the synthetic environment's generators keep their state across agents.)
Every repeat builds the agent from the same seeds, plays one rollout and runs three minibatch updates, and records SHA-1 digests of
every stage: rollout tensors, advantages, the flat gradient after each backward, the parameters after each optimiser step.  All
kernels on this path are written to be deterministic (no atomics, ordered reductions), so every digest must repeat; the first stage
whose digest differs localises a race or an uninitialised read.

Why it exists: in round 2 the two-rank test (tests/test_agent_parity2_gpu.py, two processes sharing the GPU) ended with different
parameters in two separately launched jobs in 2 of 5 full-suite runs and never in 10 runs of the test or its file alone.  That
test launches the blocking and the overlapped job as separate process pairs, so any run-to-run difference of the pipeline shows up
there as "overlapped != blocking".  This probe separates the two questions.
"""
import argparse
import hashlib
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def digest(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:12]


def poison_empty():
    """Debug aid: every torch.empty-family allocation comes back filled with NaN (floats) / a large constant (ints), so a read of
    memory nobody wrote shows up as NaN (or as an index fault) instead of as whatever the allocator handed out."""
    import torch

    def fill(t):
        if t.numel():
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype != torch.bool:
                t.fill_(0x3fffffff if t.dtype in (torch.int32, torch.int64) else 77)
        return t
    for name in ("empty", "empty_like", "empty_strided"):
        orig = getattr(torch, name)
        setattr(torch, name, (lambda o: lambda *a, **k: fill(o(*a, **k)))(orig))
    orig_new = torch.Tensor.new_empty
    torch.Tensor.new_empty = lambda self, *a, **k: fill(orig_new(self, *a, **k))


def one_run(seed, config, poison=False, dump=None):
    import torch
    if poison:
        poison_empty()
    saved = {}
    from pulse_amd import configs
    torch.manual_seed(1000)
    agent, _ = configs.make_agent(config, device="cuda:0", seed=seed, permutation_device="cpu")
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent._tensors_ready = True
    out = {}
    task = agent.vec_env.env.task
    sim = task.sim

    def env_stage(tag):
        if dump:
            saved[f"{tag}/obs_buf"] = task.obs_buf.detach().cpu().clone()
            saved[f"{tag}/rew_buf"] = task.rew_buf.detach().cpu().clone()
        out[f"{tag}/obs_buf"] = digest(task.obs_buf)
        for name in ("rigid_body_state", "dof_pos", "dof_vel"):
            if hasattr(sim, name):
                out[f"{tag}/sim.{name}"] = digest(getattr(sim, name))
        out[f"{tag}/rew_buf"] = digest(task.rew_buf)
        out[f"{tag}/progress"] = digest(task.progress_buf)
    env_stage("reset")
    task.step(torch.zeros(task.num_envs, task.num_actions, device="cuda:0"))        # one raw env step before the agent touches anything
    env_stage("step1")
    task.step(torch.zeros(task.num_envs, task.num_actions, device="cuda:0"))
    env_stage("step2")
    batch = agent.play_steps()
    batch.pop("played_frames")
    for k, v in agent.experience_buffer.tensor_dict.items():
        out[f"rollout/{k}"] = digest(v) + (f"!nan{int(torch.isnan(v).sum())}" if v.is_floating_point() and torch.isnan(v).any() else "")
    agent.set_train()
    agent.prepare_dataset(batch)
    out["advantages"] = digest(agent.dataset.values_dict["advantages"])
    agent._begin_loss_ring(3)
    for i in range(3):
        agent.train_actor_critic(agent.dataset[i])
        out[f"step{i}/grad"] = digest(agent.model.grad) + (f"!nan{int(torch.isnan(agent.model.grad).sum())}" if torch.isnan(agent.model.grad).any() else "")
        out[f"step{i}/flat"] = digest(agent.model.flat) + (f"!nan{int(torch.isnan(agent.model.flat).sum())}" if torch.isnan(agent.model.flat).any() else "")
    agent._end_loss_ring()
    torch.cuda.synchronize()
    if dump:
        torch.save(saved, dump)
    return out


def locate(dump_dir, repeats):
    """Where do the saved env-stage tensors of run 0 and the other runs differ?  Rows (envs), columns and magnitudes."""
    import torch
    runs = [torch.load(os.path.join(dump_dir, f"run{i}.pt")) for i in range(repeats)]
    for key in runs[0]:
        for i in range(1, repeats):
            a, b = runs[0][key].double(), runs[i][key].double()
            d = (a - b).abs()
            neq = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
            if not bool(neq.any()):
                bits = not torch.equal(runs[0][key].view(torch.int32), runs[i][key].view(torch.int32)) if runs[0][key].dtype == torch.float32 else False
                if bits:
                    print(f"  {key}: run 0 vs {i}: numerically equal, BIT patterns differ (signed zeros / NaN payloads)")
                continue
            if a.dim() == 2:
                rows = torch.nonzero(neq.any(dim=1)).flatten().tolist()
                cols = torch.nonzero(neq.any(dim=0)).flatten().tolist()
                print(f"  {key}: run 0 vs {i}: {int(neq.sum())} entries, max |diff| {d[neq].max().item():.3e}, rows {rows[:12]}{'...' if len(rows) > 12 else ''} "
                      f"({len(rows)}), columns {cols[:24]}{'...' if len(cols) > 24 else ''} ({len(cols)})")
            else:
                idx = torch.nonzero(neq.flatten()).flatten().tolist()
                print(f"  {key}: run 0 vs {i}: {int(neq.sum())} entries, max |diff| {d[neq].max().item():.3e}, at {idx[:24]}")


def hammer(stop):
    """A competitor process: back-to-back GEMMs on the same GPU until told to stop."""
    import torch
    from pulse_amd import kernels as K
    x = torch.randn(8192, 1024, device="cuda:0")
    w = torch.randn(2048, 1024, device="cuda:0") * 0.03
    y = torch.empty(8192, 2048, device="cuda:0")
    while not stop.is_set():
        for _ in range(50):
            K.linear_forward(x, w, out=y)
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=4)
    ap.add_argument("--config", default="cfg1")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--contend", action="store_true", help="run a second process that keeps the GPU busy with GEMMs")
    ap.add_argument("--poison", action="store_true", help="fill every torch.empty allocation with NaN: uninitialised reads become visible")
    ap.add_argument("--dump", default=None, help="directory: every repeat saves its env-stage tensors there and the differing entries are located")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-dump", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.child:
        import json
        print("DIGESTS " + json.dumps(one_run(a.seed, a.config, a.poison, a.child_dump)), flush=True)
        return 0
    proc = stop = None
    if a.contend:
        ctx = mp.get_context("spawn")
        stop = ctx.Event()
        proc = ctx.Process(target=hammer, args=(stop,))
        proc.start()
    try:
        import json
        import subprocess
        runs = []
        if a.dump:
            os.makedirs(a.dump, exist_ok=True)
        for rep in range(a.repeats):
            extra = (["--poison"] if a.poison else []) + (["--child-dump", os.path.join(a.dump, f"run{rep}.pt")] if a.dump else [])
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--config", a.config, "--seed", str(a.seed)] + extra,
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")]
            if r.returncode != 0 or not line:
                raise SystemExit(f"child failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
            runs.append(json.loads(line[0][len("DIGESTS "):]))
    finally:
        if proc is not None:
            stop.set()
            proc.join(timeout=60)
    keys = list(runs[0])
    bad = [k for k in keys if len({r[k] for r in runs}) > 1]
    for k in keys:
        vals = [r[k] for r in runs]
        print(f"{'DIFF' if k in bad else 'ok  '} {k:28s} {' '.join(vals)}")
    print(f"{len(bad)} of {len(keys)} stages differ across {a.repeats} runs" + (f"; first: {bad[0]}" if bad else ""))
    if a.dump:
        locate(a.dump, a.repeats)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
