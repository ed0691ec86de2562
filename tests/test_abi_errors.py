"""CPU: error behaviour of the C ABI (include/pulse_hip.h): invalid arguments are rejected with PULSE_ERR_INVALID_ARG and a message from
pulse_last_error() BEFORE anything touches the device, empty problems are no-ops -- the contract a foreign-language binding relies on
instead of the reference's Python exceptions.  No kernel is launched (there is no GPU in the build container)."""
import ctypes

import pytest

from pulse_amd import _lib

OK, INVALID = 0, -1


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def msg(lib):
    m = lib.pulse_last_error()
    return m.decode() if m else ""


def test_null_descriptors_are_rejected(lib):
    for fn in ("pulse_gemm_f32", "pulse_im_step", "pulse_amp_obs", "pulse_motion_state", "pulse_rollout_record", "pulse_ppo_loss"):
        assert getattr(lib, fn)(None, None) == INVALID, fn
        assert "null" in msg(lib), (fn, msg(lib))


def test_gemm_descriptor_validation(lib):
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.batch, d.split_k = 0, 128, 32, 1, 1
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == OK                      # empty problem: no-op
    d.M = -1
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "negative" in msg(lib)
    d.M = 128
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "null operand" in msg(lib)
    buf = (ctypes.c_float * 64)()
    base = (ctypes.addressof(buf) + 15) & ~15
    d.A = d.B = d.C = base
    d.lda, d.ldb, d.ldc = 30, 32, 128
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "multiples of 4" in msg(lib)
    d.lda = 32
    d.A = base + 4
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "16-byte aligned" in msg(lib)
    d.A = base
    d.a_layout, d.b_layout = _lib.GEMM_OUT_CONTIG, _lib.GEMM_RED_CONTIG
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "unsupported" in msg(lib)
    d.a_layout = d.b_layout = _lib.GEMM_RED_CONTIG
    d.rowsum = base
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "rowsum" in msg(lib)
    d.rowsum = None
    d.split_k, d.bias = 4, base
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "split-K" in msg(lib)
    d.split_k, d.bias, d.epilogue = 1, None, _lib.EPI_RELU_GRAD
    assert lib.pulse_gemm_f32(ctypes.byref(d), None) == INVALID and "aux" in msg(lib)


def test_motion_state_validation(lib):
    a = _lib.MotionStateArgs()
    a.n = 0
    assert lib.pulse_motion_state(ctypes.byref(a), None) == OK                  # no queries: no-op
    a.n = -3
    assert lib.pulse_motion_state(ctypes.byref(a), None) == INVALID
    a.n = 8
    assert lib.pulse_motion_state(ctypes.byref(a), None) == INVALID and "table" in msg(lib)


def test_rollout_record_validation(lib):
    a = _lib.RolloutRecordArgs()
    a.num_envs = 0
    assert lib.pulse_rollout_record(ctypes.byref(a), None) == OK
    a.num_envs = 16
    assert lib.pulse_rollout_record(ctypes.byref(a), None) == INVALID and "null pointer" in msg(lib)


def test_im_step_validation(lib):
    a = _lib.ImStepArgs()
    a.num_envs, a.num_bodies, a.what = 4, 24, _lib.PULSE_IM_REWARD
    assert lib.pulse_im_step(ctypes.byref(a), None) == INVALID
    assert msg(lib)                                                             # a message is always left behind


def test_version_and_sizes(lib):
    assert lib.pulse_abi_version() == _lib.ABI_VERSION
    assert lib.pulse_self_obs_width(24, 1) == 358 and lib.pulse_self_obs_width(24, 0) == 357
    assert lib.pulse_task_obs_width(6, 24, 1) == 576 and lib.pulse_task_obs_width(7, 3, 1) == 27 and lib.pulse_task_obs_width(6, 24, 3) == 1728
    assert lib.pulse_amp_obs_width(23, 4, 1) == 232 and lib.pulse_amp_obs_width(19, 4, 1) == 196 and lib.pulse_amp_obs_width(19, 4, 0) == 195
