"""Bisect the contention-dependent output of the fused env step.  Run on the GPU box:

    python tools/im_step_repro.py --children 4 --contend [--reference recorded|motion_lib] [--poison-lds]

Each child is a FRESH process that builds the cfg1 env + agent from the same seeds, then
  * saves every INPUT of the fused step (simulator tensors, recorded reference frames / motion tables, clocks) and its
    OUTPUTS (obs_buf, rew_buf, reward_raw, reset_buf, _terminate_buf) after the reset and after each of two env steps,
  * re-launches the same fused step ``--relaunch`` times on the unchanged inputs inside the process and counts launches whose
    output differs from the first launch (a race inside the kernel shows up here),
and the parent compares the children tensor by tensor: which input (if any) differs first, which output rows / columns differ
and by how much.  ``--contend`` runs a competitor process that keeps the GPU busy with GEMMs for the whole experiment.
"""
import argparse
import multiprocessing as mp
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def hammer(stop, kind="gemm"):
    """The competitor: 'gemm' = back-to-back MFMA GEMMs, 'valu' = transcendental-heavy elementwise kernels (no MFMA), 'copy' = HBM copies."""
    import torch
    if kind == "gemm":
        from pulse_amd import kernels as K
        x = torch.randn(8192, 1024, device="cuda:0")
        w = torch.randn(2048, 1024, device="cuda:0") * 0.03
        y = torch.empty(8192, 2048, device="cuda:0")
        step = lambda: K.linear_forward(x, w, out=y)
    elif kind == "valu":
        x = torch.rand(1 << 24, device="cuda:0") + 0.5
        y = torch.empty_like(x)
        step = lambda: torch.sin(torch.sqrt(x), out=y)
    else:
        x = torch.rand(1 << 26, device="cuda:0")
        y = torch.empty_like(x)
        step = lambda: y.copy_(x)
    while not stop.is_set():
        for _ in range(50):
            step()
        torch.cuda.synchronize()


def child(a):
    import torch
    from pulse_amd import configs
    from pulse_amd._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS
    torch.manual_seed(1000)
    agent, rollout = configs.make_agent(a.config, device="cuda:0", seed=a.seed, permutation_device="cpu", reference=a.reference)
    task = agent.vec_env.env.task
    sim = task.sim
    saved = {}

    def cpu(t):
        return t.detach().cpu().clone()

    def save_static():
        if rollout is not None:
            for k, v in rollout.data.items():
                saved[f"static/data.{k}"] = cpu(v)
            for name in ("ref_now", "ref_next", "ref_next_reset"):
                for k, v in getattr(rollout, name).items():
                    saved[f"static/{name}.{k}"] = cpu(v)
            saved["static/motion_lengths"] = cpu(rollout.motion_lengths)
        else:
            lib = task._motion_lib
            saved["static/frames"] = cpu(lib.frames)
            for k in ("_motion_lengths", "_motion_dt", "_motion_num_frames", "length_starts"):
                saved[f"static/{k}"] = cpu(getattr(lib, k))
            for k, v in sim.bank.items():
                saved[f"static/bank.{k}"] = cpu(v)

    def save_stage(tag):
        for name in ("rigid_body_state", "dof_pos", "dof_vel", "dof_force"):
            saved[f"{tag}/in.sim.{name}"] = cpu(getattr(sim, name))
        for name in ("progress_buf", "_pass_time", "_cycle_counter", "_motion_start_times", "_motion_start_times_offset", "_motion_len_env"):
            saved[f"{tag}/in.{name}"] = cpu(getattr(task, name))
        if task._use_motion_lib:
            saved[f"{tag}/in.sampled_motion_ids"] = cpu(task._sampled_motion_ids)
            saved[f"{tag}/in.global_offset"] = cpu(task._global_offset)
            for k, v in task._track.items():
                saved[f"{tag}/out.track.{k}"] = cpu(v)
        saved[f"{tag}/out.obs_store"] = cpu(task._obs_store)
        for name in ("rew_buf", "reward_raw", "reset_buf", "_terminate_buf"):
            saved[f"{tag}/out.{name}"] = cpu(getattr(task, name))

    def relaunch(tag, what):
        """The fused step again on the inputs it has just seen (recorded mode: nothing it reads is modified by it; motion-library mode:
        inc = 0 leaves the clock alone).  Counts launches whose outputs differ from the first one."""
        outs = ("_obs_store", "rew_buf", "reward_raw", "reset_buf", "_terminate_buf")
        first, bad, worst = None, 0, 0.0
        for i in range(a.relaunch):
            for n_ in outs:
                getattr(task, n_).fill_(7 if "buf" in n_ and getattr(task, n_).dtype == torch.int64 else 0)
            task._im_step(what)
            cur = [cpu(getattr(task, n_)) for n_ in outs]
            if first is None:
                first = cur
                continue
            diff = [not torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
                    for x, y in zip(first, cur)]
            if any(diff):
                bad += 1
                worst = max(worst, (first[0].double() - cur[0].double()).abs().nan_to_num(1e30).max().item())
        print(f"RELAUNCH {tag}: {bad} of {a.relaunch - 1} re-launches differ from the first (max |obs diff| {worst:.3e})", flush=True)
        return first

    save_static()
    agent.init_tensors()
    agent.obs = agent.env_reset()
    save_stage("reset")
    zeros = torch.zeros(task.num_envs, task.num_actions, device="cuda:0")
    for s in (1, 2):
        task.step(zeros)
        save_stage(f"step{s}")
        if not (task._use_motion_lib and task.cycle_motion):
            # outputs of a second launch on the same inputs must equal the step's own (motion-library mode: the step advanced the clock, inc = 0 now)
            first = relaunch(f"step{s}", PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS)
            same = torch.equal(first[0].view(torch.int32), saved[f"step{s}/out.obs_store"].view(torch.int32))
            print(f"RELAUNCH step{s}: first re-launch {'==' if same else '!='} the step's own observation", flush=True)
    torch.cuda.synchronize()
    torch.save(saved, a.child_dump)
    return 0


def compare(paths):
    import torch
    runs = [torch.load(p) for p in paths]
    ndiff = 0
    for key in runs[0]:
        for i in range(1, len(runs)):
            x, y = runs[0][key], runs[i][key]
            xi = x.view(torch.int32) if x.dtype == torch.float32 else x
            yi = y.view(torch.int32) if y.dtype == torch.float32 else y
            if torch.equal(xi, yi):
                continue
            ndiff += 1
            xd, yd = x.double(), y.double()
            neq = (xd != yd) & ~(torch.isnan(xd) & torch.isnan(yd))
            d = (xd - yd).abs()
            msg = f"DIFF {key}: run 0 vs {i}: {int(neq.sum())} of {neq.numel()} entries"
            if neq.any():
                msg += f", max |diff| {d[neq].nan_to_num(1e30).max().item():.3e}, nan {int(torch.isnan(xd).sum())}/{int(torch.isnan(yd).sum())}"
                if x.dim() == 2:
                    rows = torch.nonzero(neq.any(dim=1)).flatten().tolist()
                    cols = torch.nonzero(neq.any(dim=0)).flatten().tolist()
                    msg += f"; rows {rows[:16]}{'...' if len(rows) > 16 else ''} ({len(rows)}), cols {cols[:32]}{'...' if len(cols) > 32 else ''} ({len(cols)})"
                    r0 = rows[0]
                    cs = torch.nonzero(neq[r0]).flatten().tolist()[:6]
                    msg += f"; row {r0}: " + ", ".join(f"c{c}: {x[r0, c].item():.9g} vs {y[r0, c].item():.9g}" for c in cs)
            else:
                msg += " (bit patterns only: signed zero / NaN payload)"
            print(msg)
    print(f"{ndiff} differing (tensor, run) pairs over {len(runs)} runs, {len(runs[0])} tensors each")
    return ndiff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--children", type=int, default=4)
    ap.add_argument("--config", default="cfg1")
    ap.add_argument("--reference", default="recorded")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--relaunch", type=int, default=40)
    ap.add_argument("--contend", action="store_true")
    ap.add_argument("--hammer", default="gemm", choices=("gemm", "valu", "copy"))
    ap.add_argument("--lib", default=None, help="another build of libpulse_hip.so for the children (tools/build_env_step_variants.sh)")
    ap.add_argument("--no-compare", action="store_true", help="only the in-process re-launch counts")
    ap.add_argument("--poison-lds", action="store_true")
    ap.add_argument("--out", default="gpurun_out/im_repro")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-dump", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.child:
        return child(a)
    os.makedirs(a.out, exist_ok=True)
    proc = stop = None
    if a.contend:
        ctx = mp.get_context("spawn")
        stop = ctx.Event()
        proc = ctx.Process(target=hammer, args=(stop, a.hammer))
        proc.start()
    env = dict(os.environ)
    if a.poison_lds:
        env["PULSE_IM_DEBUG_POISON_LDS"] = "1"
    if a.lib:
        env["PULSE_HIP_LIB"] = os.path.abspath(a.lib)
    paths, unstable = [], 0
    try:
        for rep in range(a.children):
            path = os.path.join(a.out, f"run{rep}.pt")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--config", a.config, "--seed", str(a.seed), "--reference", a.reference,
                                "--relaunch", str(a.relaunch), "--child-dump", path], capture_output=True, text=True, timeout=900, env=env)
            print(f"--- child {rep} rc {r.returncode}")
            lines = [l for l in r.stdout.splitlines() if l.startswith("RELAUNCH")]
            print("\n".join(lines))
            unstable += sum(1 for l in lines if "!=" in l or ("re-launches differ" in l and ": 0 of " not in l))
            if r.returncode != 0:
                print(r.stdout[-3000:], r.stderr[-3000:])
                return 2
            paths.append(path)
    finally:
        if proc is not None:
            stop.set()
            proc.join(timeout=60)
    n = 0 if a.no_compare else compare(paths)
    for p in paths:
        os.remove(p)
    print(f"{unstable} unstable re-launch series")
    return 1 if (n or unstable) else 0


if __name__ == "__main__":
    raise SystemExit(main())
