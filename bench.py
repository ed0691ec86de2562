"""Headline benchmark: env-steps/sec through the PPO update (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one ``agent.train_epoch()`` over one synthetic rollout batch: T=32 env steps of the fused
observation / imitation-reward / reset kernel + actor/critic inference + experience-buffer writes for
N=4096 SMPL humanoids, GAE, dataset prep, and 6 x 8 PPO minibatch steps (forward, losses, backward,
[RCCL all-reduce], clip, Adam) on the [1024, 512] actor + critic -- BASELINE.json configs[1].
All inputs are resident in HBM before the timed region.  With N > 1 every rank owns its own 4096
environments (weak scaling, envs shard with no data-path collective; the one exchange per optimiser
step is the flat-gradient all-reduce).

Prints ONE JSON line on rank 0.  ``roofline`` is measured live: every launch of the dominant
kernel (gemm_x3_kernel: the fp32 GEMM computed as six bf16 MFMAs per k step over an exact three-way operand split; or
gemm_f32_kernel, the fp32 MFMA GEMM, with PULSE_GEMM_F32=mfma32) inside the LAST step of the timed region is bracketed by
a HIP event pair on the launch stream; achieved = algorithmic FLOPs / summed kernel time.  ``cpu_baseline`` is
the PyTorch-CPU oracle agent (oracle/agent_oracle.py, kind "port") timed on the host cores on a
bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
MFMA_X3_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0   # fp32 GEMM as 6 bf16 MFMAs per algorithmic product group: its matrix-pipe ceiling


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reference", default="motion_lib", choices=["motion_lib", "recorded"],
                    help="reference-motion source: HBM-resident motion library queried every step, or pre-recorded frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-minibatches", type=int, default=4, help="PPO minibatch steps timed on the CPU oracle")
    ap.add_argument("--cpu-steps", type=int, default=8, help="rollout steps timed on the CPU oracle")
    ap.add_argument("--cpu-budget", type=float, default=240.0, help="wall-clock bound of the CPU baseline subprocess (s)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true", help="do not bracket GEMM launches with events")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the sustained-clock probe (60 extra launches of the layer-1 forward GEMM after the timed region)")
    ap.add_argument("--keep-gc", action="store_true", help="leave Python's cyclic garbage collector running during the timed region")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns the config's env count; strong: the config's env count is divided over the ranks")
    return ap.parse_args()


def cpu_baseline_worker(cfg_name, seed, sample_steps, n_minibatches, reference="motion_lib"):
    """Runs in a SUBPROCESS (hard wall-clock bound): the oracle agent on the host cores for a bounded sample -- ``sample_steps`` of
    the T rollout steps and ``n_minibatches`` (>= 3) PPO minibatch steps at the full N and the full minibatch size.  BASELINE.md
    protocol: the intra-op thread count is swept once on a minibatch step (oversubscribing a 256-thread host costs 3x), the best
    count is used, the rollout step time is the mean of the sampled steps and the minibatch time the MEDIAN of the repetitions
    after one warm-up; both are extrapolated linearly to the epoch."""
    import statistics
    import torch
    from oracle import agent_oracle as AO
    from pulse_amd import configs, synthetic as syn
    from pulse_amd.env.sim import RecordedRollout
    cfg, num_envs = configs.agent_config(cfg_name)
    T = cfg["horizon_length"]
    total_mb = cfg["mini_epochs"] * (T * num_envs // cfg["minibatch_size"])
    scfg = dict(cfg)
    scfg["horizon_length"] = sample_steps
    assert sample_steps * num_envs % cfg["minibatch_size"] == 0
    torch.manual_seed(seed)
    if reference == "motion_lib":
        from oracle import motion_oracle as MO
        env = MO.make_agent_env(num_envs, sample_steps, seed)
    else:
        rollout = RecordedRollout(num_envs, sample_steps + 1, seed=seed)
        env = AO.OracleEnv(rollout, syn.RESET_BODY_IDS, list(range(24)))
    agent = AO.OracleCommonAgent(scfg, env, cfg["network"]["mlp"]["units"], seed=seed)
    agent.obs = env.reset()
    t_all = time.time()
    # ---- thread sweep on one minibatch step of synthetic rows (shape of the real thing)
    logical = os.cpu_count() or 1
    mb = cfg["minibatch_size"]
    probe = {"obs": torch.randn(mb, 934), "actions": torch.randn(mb, 69) * 0.1, "old_logp_actions": torch.randn(mb), "advantages": torch.randn(mb),
             "old_values": torch.randn(mb, 1), "returns": torch.randn(mb, 1), "mu": torch.randn(mb, 69) * 0.1, "sigma": torch.full((mb, 69), 0.055)}
    sweep = {}
    for th in [t for t in (8, 16, 32, 64, 128, 256) if t <= logical] or [logical]:
        torch.set_num_threads(th)
        agent.calc_gradients(probe)                               # warm-up (thread pool, allocator); timing only, the weights do not matter
        t0 = time.time()
        agent.calc_gradients(probe)
        sweep[th] = time.time() - t0
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    # ---- rollout sample
    t0 = time.time()
    with torch.no_grad():
        batch = agent.play_steps()
    play = time.time() - t0
    agent.set_train()
    agent.prepare_dataset(batch)
    nmb = sample_steps * num_envs // mb
    agent.calc_gradients(agent._get_item(0))                      # warm-up
    times = []
    for i in range(max(3, n_minibatches)):
        t0 = time.time()
        agent.calc_gradients(agent._get_item(i % nmb))
        times.append(time.time() - t0)
    per_step = play / sample_steps
    per_mb = statistics.median(times)
    epoch_s = per_step * T + per_mb * total_mb
    wall = time.time() - t_all
    return {"value": T * num_envs / epoch_s, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cfg_name} ({reference} reference) at full width ({num_envs} envs, minibatch {mb}): {sample_steps} of {T} rollout steps "
                      f"({play:.1f} s) + median of {len(times)} PPO minibatch steps after one warm-up ({per_mb:.2f} s each; {total_mb} per epoch), "
                      f"extrapolated linearly to the epoch; intra-op threads swept {sweep} -> {cores} of {logical} logical CPUs; "
                      f"{wall:.1f} s of CPU work; oracle/agent_oracle.py, torch {torch.__version__} eager fp32",
            "rollout_s_per_step": per_step, "update_s_per_minibatch": per_mb, "epoch_s_extrapolated": epoch_s,
            "thread_sweep_s_per_minibatch": {str(k): v for k, v in sweep.items()}}


def gemm_traffic(cfg_name, kernel):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate passes, mean over the GEMM launches of one epoch of this config); None if not collected.  The
    counters serialise kernels, so they cannot be read inside the timed region: the number is a property of the same
    command profiled once per round."""
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r*_gemm_traffic_{cfg_name}.json")))
    for path in reversed(paths):              # newest round first
        try:
            with open(path) as f:
                d = json.load(f)
            if kernel in d:
                return d[kernel]["hbm_bytes_per_launch"], os.path.join("profiles", os.path.basename(path))
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def gemm_clock_probe(m, n, k, launches=30):
    """Sustained shader clock under the dominant GEMM: every workgroup of ``launches`` back-to-back layer-1-forward launches stamps
    s_memtime (shader cycles) and the 100 MHz wall clock at its start and end (pulse_gemm_set_debug_buffer); the ratio over the
    last launch's workgroups is the clock the chip actually held.  MI355X does NOT hold 2.4 GHz under dense MFMA load (power
    management), so ``roofline.frac`` against the 2.4 GHz peak mixes kernel efficiency with the clock; this number separates them."""
    import torch
    from pulse_amd import _lib, kernels
    dev = torch.device("cuda", torch.cuda.current_device())
    kp = (k + 31) // 32 * 32
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(m, kp, device=dev, generator=g).clamp_(min=0)          # activations as the network sees them (half zeros)
    w = torch.randn(n, kp, device=dev, generator=g) * 0.05
    out = torch.empty(m, n, device=dev)
    nwg = ((m + 127) // 128) * ((n + 127) // 128)
    stamps = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    lib = _lib.load()
    for _ in range(launches):
        kernels.gemm(x, w, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=n, activation=_lib.ACT_RELU)
    torch.cuda.synchronize()
    _lib.check(lib.pulse_gemm_set_debug_buffer(stamps.data_ptr()), "pulse_gemm_set_debug_buffer")
    try:
        for _ in range(launches):
            kernels.gemm(x, w, out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=n, activation=_lib.ACT_RELU)
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.pulse_gemm_set_debug_buffer(None), "pulse_gemm_set_debug_buffer")
    # the buffer is sized for the 128 x 128 tiling (its upper bound); the launcher may have taken the 256 x 256 tile for this shape, whose
    # workgroups stamp the first rows only -- the tiling that ran is reported beside the clock
    tile = int(lib.pulse_gemm_last_tile())
    used = ((m + tile - 1) // tile) * ((n + max(tile, 128) - 1) // max(tile, 128)) if tile in (64, 128, 256) else nwg
    st = stamps[:used].cpu().double()
    cyc, wall = (st[:, 4] - st[:, 0]).sum().item(), (st[:, 5] - st[:, 1]).sum().item()
    gemm_clock_probe.tile = tile
    return cyc / (wall * 10.0) if wall > 0 else None


def cpu_baseline(cfg_name, seed, sample_steps, n_minibatches, budget_s, reference="motion_lib"):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--config", cfg_name, "--cpu-steps", str(sample_steps),
           "--cpu-minibatches", str(n_minibatches), "--reference", reference]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""                           # the worker never touches the GPU
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=budget_s, env=env)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"value": None, "unit": "env-steps/s", "cores": None, "kind": "port", "sample": f"worker failed: {p.stderr[-300:]}"}
        return json.loads(line[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "env-steps/s", "cores": None, "kind": "port", "sample": f"worker exceeded its {budget_s} s budget"}


def workload_suffix(cfg):
    """The parts of a config beyond plain PPO, spelled out in config.workload (BASELINE.json configs[2] / [4])."""
    parts = []
    net = cfg["network"]
    if cfg.get("enable_disc"):
        parts.append(f"AMP discriminator MLP {net['disc']['units']} on 10-step AMP observations (amp_minibatch {cfg['amp_minibatch_size']}, "
                     f"demo + replay rings {cfg['amp_obs_demo_buffer_size']} rows, disc reward mixed {cfg['disc_reward_w']}/{cfg['task_reward_w']})")
    if net.get("name") == "amp_z":
        parts.append(f"PULSE VAE encoder/decoder policy head (network amp_z, latent 32, task MLP {net['task_mlp']['units']})"
                     + (", distillation (kin) loss only" if cfg.get("_env_kind") == "vae" else ""))
    if cfg.get("mixed_precision"):
        parts.append("mixed_precision: training GEMMs on the bf16 MFMA over fp32 master weights")
    return (" + " + " + ".join(parts)) if parts else ""


def launch_plan(gpus, env):
    """What a `bench.py --gpus N` process has to do, from its arguments and environment alone (no torch; unit-tested on CPU).

    The reference turns multi-GPU on with one config flag (im.yaml:49 `multi_gpu`, common_agent.py:112-127) and expects one
    process per GPU to exist already (horovodrun); the driver's command is plain `python bench.py --gpus N`.  So:
      * WORLD_SIZE set (a launcher started us): it must equal --gpus, otherwise refuse -- never print n_gpus != --gpus;
      * WORLD_SIZE unset and --gpus 1: run in this process;
      * WORLD_SIZE unset and --gpus N > 1: this process becomes the launcher of N rank processes (self_launch)."""
    if gpus < 1:
        return {"action": "refuse", "why": f"--gpus {gpus}"}
    ws = env.get("WORLD_SIZE")
    if ws is None:
        if gpus == 1:
            return {"action": "run", "world": 1, "rank": 0, "local_rank": 0}
        return {"action": "spawn", "world": gpus}
    world = int(ws)
    if world != gpus:
        return {"action": "refuse", "why": f"--gpus {gpus} but WORLD_SIZE={world}: refusing to report a rank count that is not the one asked for"}
    rank = int(env.get("RANK", "0"))
    if not 0 <= rank < world:
        return {"action": "refuse", "why": f"RANK={rank} outside WORLD_SIZE={world}"}
    return {"action": "run", "world": world, "rank": rank, "local_rank": int(env.get("LOCAL_RANK", str(rank)))}


def rank_environment(base_env, rank, world, port):
    """Environment of rank ``rank`` of a self-launched job: what torch.distributed.run would export (127.0.0.1 rendezvous: the
    container hostname may not resolve), dmabuf IPC for RCCL, and one OpenMP thread per rank (N ranks share the host)."""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def self_launch(gpus, argv):
    """`python bench.py --gpus N` without a launcher: start one rank process per GPU (same interpreter, same arguments), rank 0's
    stdout (the ONE JSON line) is this process's stdout, every rank's stderr is forwarded.  Returns the job's exit code: the first
    non-zero rank code (the other ranks are then terminated -- a dead rank would leave them in a collective forever)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    log(f"self-launch: {gpus} ranks, rendezvous 127.0.0.1:{port}")
    procs = []
    for r in range(gpus):
        procs.append(subprocess.Popen(cmd, env=rank_environment(os.environ, r, gpus, port),
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=None))
    rc = 0
    pending = set(range(gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                log(f"self-launch: rank {r} exited with {code}; stopping the other ranks")
                for q in pending:
                    procs[q].terminate()                          # exact PIDs we started
        time.sleep(0.05)
    return rc


def main():
    a = parse()
    if a.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(a.config, 1234, a.cpu_steps, a.cpu_minibatches, a.reference)), flush=True)
        return
    plan = launch_plan(a.gpus, os.environ)
    if plan["action"] == "refuse":
        raise SystemExit("bench.py: " + plan["why"])
    if plan["action"] == "spawn":
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    world, rank, local_rank = plan["world"], plan["rank"], plan["local_rank"]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    share = bool(os.environ.get("PULSE_BENCH_SHARE_GPU"))
    if torch.cuda.is_available() and torch.cuda.device_count() < world and not share:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) visible "
                         "(set PULSE_BENCH_SHARE_GPU=1 with PULSE_DIST_BACKEND=gloo for a functional run of the N>1 path on one device)")
    from pulse_amd import _lib, configs, kernels
    from pulse_amd.env.sim import RecordedRollout
    from pulse_amd.parallel import DistContext
    _lib.load()                                                   # fail loudly before anything else if the HIP library is missing
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if share:                                                     # functional test of the N>1 path on a 1-GPU box (with PULSE_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    dist = DistContext(enabled=world > 1)
    seed = 1234
    cfg, num_envs = configs.agent_config(a.config)
    T = cfg["horizon_length"]
    over = {}
    if a.scaling == "strong" and world > 1:
        assert num_envs % world == 0
        num_envs //= world                                        # the SAME total work split over the ranks
        over = {"num_envs_override": num_envs, "minibatch_size": max(cfg["minibatch_size"] // world, num_envs)}
    # synthetic inputs, seed 1234 + rank: a motion library resident in HBM (+ tracking physics stand-in) or recorded frames
    rollout_cpu = RecordedRollout(num_envs, T + 1, seed=seed, rank=rank) if a.reference == "recorded" else None
    agent, _ = configs.make_agent(a.config, device=device, seed=seed, rank=rank, rollout=rollout_cpu, reference=a.reference,
                                  multi_gpu=world > 1, dist=dist, **over)
    log(f"rank {rank}: synthetic inputs generated ({a.reference})")
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent._tensors_ready = True
    if world > 1:
        dist.setup_algo(agent.model.flat, (agent.model.sigma, agent.exp_avg, agent.exp_avg_sq))

    if not a.keep_gc:
        # A generation-2 collection walks every live object (hundreds of thousands here: tensors, ctypes descriptors, plans) and
        # stalls the launch thread for tens of ms -- long enough for the GPU queue to run dry.  The training loop creates no
        # reference cycles that matter, so the collector is parked for the run (reference counting still frees everything).
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
    log(f"rank {rank}: agent ready, warmup")
    for _ in range(a.warmup):
        agent.train_epoch()
    log(f"rank {rank}: timing {a.steps} steps")
    prof = kernels.PROFILER
    dist.time_exposure = world > 1                                # event-bracket every wait for gradient buckets (device-side exposure)
    dist.exposed_wait_ms()                                        # (drop what the warm-up recorded)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    play = upd = 0.0
    per_step = []
    for step in range(a.steps):
        if step == a.steps - 1 and not a.no_roofline:
            prof.start()      # the LAST timed step carries the HIP-event brackets (an event pair costs ~10 us of stream time per GEMM)
        info = agent.train_epoch()
        play += info["play_time"]
        upd += info["update_time"]
        per_step.append((round(1e3 * info["play_time"], 2), round(1e3 * info["update_time"], 2)))
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof.stop()
    elapsed = dist.max_over_ranks(elapsed)
    log(f"rank {rank}: timed region done: {elapsed:.3f} s")

    env_steps = a.steps * T * num_envs * world
    out = {
        "metric": "env-steps/sec through PPO update", "value": env_steps / elapsed, "unit": "env-steps/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": a.scaling,
        "vs_baseline": None, "dtype": "bf16" if cfg.get("mixed_precision") else "f32", "data": "synthetic",
        # BASELINE.json configs[3] ("32768 envs sharded 8-way ...") is this workload at 8 ranks: name it when the run is that shape
        "config": {"workload": ("cfg4 (cfg2 sharded over %d GPUs, %d envs in total): " % (world, num_envs * world) if (a.config == "cfg2" and world > 1) else "") +
                               f"{a.config}: {num_envs} SMPL-humanoid envs/GPU x horizon {T}, imitation obs/reward/reset + PPO "
                               f"(actor+critic MLP {cfg['network']['mlp']['units']}, minibatch {cfg['minibatch_size']} x {cfg['mini_epochs']} mini-epochs)"
                               + workload_suffix(cfg),
                   "num_envs_per_gpu": num_envs, "horizon": T, "global_batch": T * num_envs * world, "parallelism": f"dp{world}",
                   "reference_motion": "HBM-resident motion library (1024 clips), queried every step" if a.reference == "motion_lib"
                   else "pre-recorded reference frames"},
        "play_ms_per_step": 1e3 * play / a.steps, "update_ms_per_step": 1e3 * upd / a.steps, "per_step_play_update_ms": per_step,
    }
    if out["n_gpus"] != a.gpus:
        raise SystemExit(f"bench.py: ran {out['n_gpus']} rank(s) for --gpus {a.gpus}")
    if world > 1:
        ids = [None] * world
        import torch.distributed as tdist
        tdist.all_gather_object(ids, {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
                                      "uuid": str(getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "uuid", ""))})
        if dist.backend_world_size() != world:
            raise SystemExit(f"bench.py: backend came up with {dist.backend_world_size()} ranks, --gpus {world}")
        st = dist.stats()
        exp_ms, exp_n = dist.exposed_wait_ms()
        out["allreduce"] = {"backend": st["backend"], "ranks": world, "backend_world_size": dist.backend_world_size(), "rank_devices": ids,
                            "devices_shared": share,
                            "calls_per_step": st["calls"] / max(1, a.steps + a.warmup),
                            "mbytes_per_call": st["bytes"] / max(1, st["calls"]) / 1e6,
                            "ms_per_step_host_enqueue": 1e3 * st["seconds"] / max(1, a.steps + a.warmup),
                            # device-side: how long the optimiser's stream actually waited for gradient buckets (what the overlap did not hide)
                            "ms_per_step_exposed_wait_device": exp_ms / max(1, a.steps), "waits_timed": exp_n}
    if not a.no_roofline:
        s = prof.summary()

        def roof(tags, peak, kernel, traffic_key=None):
            sel = {k: v for k, v in s.items() if k in tags}
            n = sum(v[0] for v in sel.values())
            t = sum(v[1] for v in sel.values())
            f = sum(v[2] for v in sel.values())
            if n == 0 or t == 0:
                return None
            traffic, traffic_src = gemm_traffic(a.config, traffic_key or kernel)
            return {"bound": "mfma", "achieved": f / t / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": f / t / 1e12 / peak,
                    # NOT measured in this run (the counters serialise kernels): the newest committed PMC pass of the same command, named here
                    "traffic": traffic, "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes, of the same bench command)") if traffic_src else None,
                    "kernel": kernel, "launches": n, "avg_us": 1e6 * t / max(1, n),
                    "kernel_time_frac_of_step": t / (elapsed / a.steps), "instrumented_steps": 1,
                    "by_variant": {k: {"launches": v[0], "avg_us": 1e6 * v[1] / v[0], "tflops": v[2] / v[1] / 1e12} for k, v in sel.items()}}
        r32 = roof(("fwd", "dx", "dw"), MFMA_F32_PEAK_TFLOPS, "gemm_f32_kernel")
        if r32 is None:
            # the dominant kernel of the step is the 256 x 256 tile of the x3 GEMM (gemm_x3w_kernel) when the launcher took it for most of the
            # GEMM time: ``roofline`` is THAT kernel's (its launches only); ``all_fp32_gemm_launches`` beside it covers every x3 launch of the step
            # -- both tilings, the skinny head GEMMs and the rollout's M = num_envs launches included -- the figure rounds 2-4 reported
            allx3 = roof(("x3_fwd", "x3_dx", "x3_dw", "x3p_fwd", "x3w_fwd", "x3w_dx", "x3w_dw"), MFMA_X3_PEAK_TFLOPS,
                         "every x3 fp32 GEMM launch (256 x 256 tile: gemm_x3w_kernel; 128 x 128 / 64 x 128: gemm_x3_kernel)", traffic_key="gemm_x3_kernel")
            wide = roof(("x3w_fwd", "x3w_dx", "x3w_dw"), MFMA_X3_PEAK_TFLOPS, "gemm_x3w_kernel", traffic_key="gemm_x3w_kernel")
            if wide is not None and wide["traffic"] is None:          # no PMC pass that separates the tilings yet: the mixed figure, named as such
                wide["traffic"], src = gemm_traffic(a.config, "gemm_x3_kernel")
                wide["traffic_source"] = (src + " (mean over the launches of BOTH tilings: that pass does not separate them)") if src else None
            narrow = roof(("x3_fwd", "x3_dx", "x3_dw", "x3p_fwd"), MFMA_X3_PEAK_TFLOPS, "gemm_x3_kernel", traffic_key="gemm_x3_kernel")
            if wide is not None and (narrow is None or wide["launches"] * wide["avg_us"] >= narrow["launches"] * narrow["avg_us"]):
                r32 = wide
                # SCOPE (round-5 advisor finding): ``roofline`` is the DOMINANT KERNEL's figure, as the bench contract asks; rounds 2-4 reported every
                # fp32 GEMM launch -- that figure, comparable across rounds, is ``all_fp32_gemm_launches``
                r32["scope"] = "launches of the dominant kernel (gemm_x3w_kernel) only; all_fp32_gemm_launches = every x3 launch of the step (the figure of rounds 2-4)"
                r32["all_fp32_gemm_launches"] = {k: allx3[k] for k in ("achieved", "frac", "launches", "avg_us", "kernel_time_frac_of_step", "by_variant")}
            else:
                r32 = allx3
            if r32 is not None:
                r32["arithmetic"] = ("fp32 in / fp32 out; operands split exactly into 3 bf16 planes, 6 v_mfma_f32_32x32x16_bf16 per 16-deep k step, "
                                     "fp32 accumulation; peak = dense bf16 MFMA peak / 6; the fp32 MFMA's own ceiling is %.1f TFLOP/s" % MFMA_F32_PEAK_TFLOPS)
                # not part of the contract's frac: what the fully issued pipe sustains on random operands with nothing to feed (it is clocked to
                # 1.72 GHz by operand toggling alone), measured once with tools/mfma_ceiling_probe.cpp -> profiles/r03_mfma_ceiling.txt
                r32["pipe_only_ceiling"] = {"tflops": 1758.3 / 6, "frac_of_it": r32["achieved"] / (1758.3 / 6),
                                            "source": "profiles/r03_mfma_ceiling.txt (register-resident MFMA chains, random bf16 operands: 1758.3 TFLOP/s at 1.72 GHz; "
                                                      "2474.5 at 2.39 GHz on all-zero operands)"}
        r16 = roof(("b16_fwd", "b16_dx", "b16_dw"), MFMA_BF16_PEAK_TFLOPS, "bf16-storage GEMMs (gemm_b16r_kernel, gemm_b16w_kernel, gemm_x3p_kernel<.., 1>)")
        if r16 is None:
            r16 = roof(("bf16_fwd", "bf16_dx", "bf16_dw"), MFMA_BF16_PEAK_TFLOPS, "gemm_bf16_kernel")
        if r16 is not None and getattr(agent, "_disc_stream", None) is not None:
            # the discriminator chain runs on its own stream beside the actor / critic chain: an event pair around a launch then spans whatever
            # the other chain had on the chip at the time, so avg_us / achieved are in-situ figures UNDER CONCURRENCY (rocprofv3's kernel
            # durations of the same command see the same); PULSE_DISC_STREAM=0 gives the one-chain-at-a-time durations
            r16["concurrent_chains"] = True
        if r16 is not None:
            # mixed precision: the training GEMMs run on the bf16 MFMA (judged against its 2.5 PFLOP/s dense peak; with fp32 operand
            # storage the kernel is bound by operand traffic); the rollout's fp32 inference GEMMs are reported beside it
            out["roofline"] = r16
            if r32 is not None:
                out["roofline_fp32_gemm"] = r32
        elif r32 is not None:
            out["roofline"] = r32
        if rank == 0 and "roofline" in out and not a.no_clock_probe:
            try:
                net = getattr(agent.model, "a2c_network", agent.model)
                units = getattr(net, "units", None) or [1024]
                ghz = gemm_clock_probe(int(agent.minibatch_size), 2 * int(units[0]), int(agent.obs_shape[0]))
            except Exception as e:                                   # the probe is diagnostics: never lose the bench line over it
                log(f"clock probe failed: {e}")
                ghz = None
            if ghz:
                r = out["roofline"]
                r["sustained_clock_ghz"] = ghz
                r["frac_at_sustained_clock"] = r["achieved"] / (r["peak"] * ghz / 2.4)
                r["clock_probe_tile"] = getattr(gemm_clock_probe, "tile", None)     # rows of the tile the probe's launches ran on (256 = gemm_x3w_kernel)
                r["clock_note"] = ("peak is quoted at 2.4 GHz; sustained_clock_ghz is the shader clock measured (per-workgroup s_memtime vs wall stamps) "
                                   "under 30 back-to-back launches of this config's layer-1 forward GEMM right after the timed region")
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("timing the CPU oracle (subprocess, bounded)")
        out["cpu_baseline"] = cpu_baseline(a.config, seed, a.cpu_steps, a.cpu_minibatches, a.cpu_budget, a.reference)
        if out["cpu_baseline"]["value"]:
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.shutdown()


if __name__ == "__main__":
    main()
