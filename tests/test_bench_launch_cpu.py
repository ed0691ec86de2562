"""bench.py's launch logic (no GPU, no torch.distributed): `python bench.py --gpus N` must never report a rank count other
than N -- it becomes the launcher of N rank processes when no launcher started it (the reference expects horovodrun to have
done that: common_agent.py:112-127)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("pulse_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_plan():
    b = _bench()
    assert b.launch_plan(1, {}) == {"action": "run", "world": 1, "rank": 0, "local_rank": 0}
    assert b.launch_plan(8, {}) == {"action": "spawn", "world": 8}                     # no launcher: start the ranks ourselves
    p = b.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"})
    assert p == {"action": "run", "world": 4, "rank": 3, "local_rank": 3}
    assert b.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "2"})["local_rank"] == 2       # LOCAL_RANK defaults to RANK (one node)
    assert b.launch_plan(8, {"WORLD_SIZE": "2", "RANK": "0"})["action"] == "refuse"    # torchrun with another count
    assert b.launch_plan(1, {"WORLD_SIZE": "2", "RANK": "0"})["action"] == "refuse"
    assert b.launch_plan(2, {"WORLD_SIZE": "1", "RANK": "0"})["action"] == "refuse"    # the silent 1-GPU run of round 3
    assert b.launch_plan(2, {"WORLD_SIZE": "2", "RANK": "2"})["action"] == "refuse"
    assert b.launch_plan(0, {})["action"] == "refuse"


def test_rank_environment():
    b = _bench()
    e = b.rank_environment({"PATH": "/x", "OMP_NUM_THREADS": "4"}, 3, 8, 12345)
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "12345")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["OMP_NUM_THREADS"] == "4" and e["PATH"] == "/x"
    assert b.launch_plan(8, e) == {"action": "run", "world": 8, "rank": 3, "local_rank": 3}


def test_workload_names_the_discriminator_and_the_vae():
    b = _bench()
    from pulse_amd import configs
    c5, _ = configs.agent_config("cfg5")
    s5 = b.workload_suffix(c5)
    assert "AMP discriminator MLP [1024, 512]" in s5 and "bf16" in s5
    c3, _ = configs.agent_config("cfg3")
    assert "PULSE VAE encoder/decoder" in b.workload_suffix(c3)
    c2, _ = configs.agent_config("cfg2")
    assert b.workload_suffix(c2) == ""


def test_refuses_world_size_mismatch_with_nonzero_rc():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


def test_self_launch_propagates_rank_failure():
    """No GPU here: every self-launched rank fails in its first lines (no HIP device); the launcher must return non-zero and print no JSON."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "self-launch: 2 ranks" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
