"""Run-to-run reproducibility of the hot path beside a COMPETING process (round-2 verdict, weak #1).

Round 2 found the fused env step's observation row differing from launch to launch on identical inputs whenever a second process
ran GEMMs on the same GPU.  Round 3 bisected it (tools/im_step_repro.py, tools/pk_f32_probe.cpp): packed-fp32 VALU instructions
(v_pk_mul_f32 / v_pk_add_f32, what the SLP vectoriser turns adjacent scalar fp32 math into) return wrong values in the last
quarter of the wave while another kernel's waves issue MFMAs on the same SIMD.  pulse_amd/csrc/build.py now builds every
translation unit without them and audits the device code.  These tests pin the behaviour: a GEMM-hammering competitor process, the
fused step re-launched on unchanged inputs, fresh processes compared tensor by tensor.
"""
import multiprocessing as mp
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _hammer(stop, ready):
    import torch as th
    from pulse_amd import kernels as K
    x = th.randn(8192, 1024, device="cuda:0")
    w = th.randn(2048, 1024, device="cuda:0") * 0.03
    y = th.empty(8192, 2048, device="cuda:0")
    K.linear_forward(x, w, out=y)
    th.cuda.synchronize()
    ready.set()
    while not stop.is_set():
        for _ in range(50):
            K.linear_forward(x, w, out=y)
        th.cuda.synchronize()


@pytest.fixture
def gemm_competitor():
    ctx = mp.get_context("spawn")
    stop, ready = ctx.Event(), ctx.Event()
    proc = ctx.Process(target=_hammer, args=(stop, ready))
    proc.start()

    def start_and_wait():
        assert ready.wait(timeout=300), "the competitor process did not come up"
    yield start_and_wait
    stop.set()
    proc.join(timeout=60)
    if proc.is_alive():
        proc.terminate()


@pytest.mark.parametrize("reference", ["recorded", "motion_lib"])
def test_fused_env_step_bit_stable_beside_gemm_process(gemm_competitor, reference):
    from pulse_amd import configs
    from pulse_amd._lib import PULSE_IM_RESET, PULSE_IM_REWARD, PULSE_IM_SELF_OBS, PULSE_IM_TASK_OBS
    agent, _ = configs.make_agent("cfg1", device="cuda:0", seed=7, reference=reference)
    task = agent.vec_env.env.task
    agent.init_tensors()
    agent.env_reset()
    task.step(torch.zeros(task.num_envs, task.num_actions, device="cuda:0"))
    outs = ("_obs_store", "rew_buf", "reward_raw", "reset_buf", "_terminate_buf")
    what = PULSE_IM_REWARD | PULSE_IM_RESET | PULSE_IM_SELF_OBS | PULSE_IM_TASK_OBS

    def launch():
        for n in outs:
            getattr(task, n).zero_()
        task._im_step(what)                     # same inputs: recorded mode reads nothing it writes, library mode keeps the clock (inc = 0)
        torch.cuda.synchronize()
        return [getattr(task, n).clone() for n in outs]
    want = launch()                              # the competitor may or may not be up yet: the result must not depend on it
    gemm_competitor()
    for i in range(60):
        got = launch()
        for n, a, b in zip(outs, want, got):
            same = torch.equal(a.view(torch.int32), b.view(torch.int32)) if a.dtype == torch.float32 else torch.equal(a, b)
            assert same, f"launch {i}: {n} differs from the first launch (max |diff| {(a.double() - b.double()).abs().max().item():.3e})"


def test_fresh_processes_agree_beside_gemm_process(tmp_path):
    """Two fresh processes per reference source, every input and output tensor of reset + two env steps compared bit for bit, and 12
    in-process re-launches each -- all beside a GEMM-hammering third process (tools/im_step_repro.py exits non-zero on any difference)."""
    for reference in ("recorded", "motion_lib"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "im_step_repro.py"), "--children", "2", "--contend", "--relaunch", "12",
                            "--reference", reference, "--out", str(tmp_path / reference)], capture_output=True, text=True, timeout=900, cwd=ROOT)
        tail = "\n".join(r.stdout.splitlines()[-12:])
        assert r.returncode == 0, f"{reference}: run-to-run differences beside a competitor process:\n{tail}\n{r.stderr[-1500:]}"
        assert "0 unstable re-launch series" in r.stdout and " 0 of 11 re-launches differ" in r.stdout, tail
