"""`python bench.py --gpus 2` with NO launcher, on the one GPU of the test box: the two self-launched ranks share the device
(PULSE_BENCH_SHARE_GPU) and talk over gloo (PULSE_DIST_BACKEND): the N > 1 control flow end to end, the JSON line reporting the
rank count that was asked for (round-3 verdict: without torchrun the bench silently ran one rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, extra_env, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_launches_two_ranks():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-clock-probe"],
             {"PULSE_BENCH_SHARE_GPU": "1", "PULSE_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                   # rank 0 prints the ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0 and d["scaling"] == "weak"
    ar = d["allreduce"]
    assert ar["backend"] == "gloo" and ar["backend_world_size"] == 2 and ar["ranks"] == 2 and ar["devices_shared"] is True
    assert [r["rank"] for r in ar["rank_devices"]] == [0, 1]
    assert "cfg4" in d["config"]["workload"] and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 2 * 4096 * 32
    assert d["value"] > 0 and "roofline" in d


def test_bench_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count() + 1
    p = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "device(s) visible" in p.stderr
