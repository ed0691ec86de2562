"""CPU: the C-ABI library loads and exports every symbol include/pulse_hip.h declares (no compute)."""
import ctypes
import os
import re

from pulse_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pulse_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pulse_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_header_symbols():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in pulse_hip.h but not exported"
    assert lib.pulse_abi_version() == _lib.ABI_VERSION


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_host_side_helpers_without_gpu():
    lib = _lib.load()
    assert lib.pulse_self_obs_width(24, 1) == 358          # humanoid.py:653
    assert lib.pulse_task_obs_width(6, 24, 1) == 576       # humanoid_im.py:476-477
    assert lib.pulse_task_obs_width(7, 3, 1) == 27
    assert lib.pulse_task_obs_width(6, 24, 3) == 3 * 576


def test_invalid_arguments_are_reported_not_thrown():
    lib = _lib.load()
    assert lib.pulse_im_step(None, None) == -1
    assert b"null args" in lib.pulse_last_error()
    assert lib.pulse_quat_mul(None, None, None, -5, None) == -1
    # zero-sized work is a no-op even with null pointers
    assert lib.pulse_quat_mul(None, None, None, 0, None) == 0
    assert lib.pulse_gae(None, None, None, None, 0, 0, 0, 0, 0.99, 0.94, None, None, None) == 0


def test_struct_layout_matches_c():
    # sizeof(pulse_im_step_args) as the C compiler lays it out (natural alignment, LP64)
    assert ctypes.sizeof(_lib.RewardSpecs) == 40
    assert ctypes.sizeof(_lib.ImStepArgs) == _lib.load().pulse_sizeof_im_step_args()
    assert ctypes.sizeof(_lib.GemmDesc) == _lib.load().pulse_sizeof_gemm_desc()
    assert ctypes.sizeof(_lib.AmpObsArgs) == _lib.load().pulse_sizeof_amp_obs_args()
    assert ctypes.sizeof(_lib.PpoLossArgs) == _lib.load().pulse_sizeof_ppo_loss_args()
    assert ctypes.sizeof(_lib.MotionStateArgs) == _lib.load().pulse_sizeof_motion_state_args()
    assert ctypes.sizeof(_lib.RolloutRecordArgs) == _lib.load().pulse_sizeof_rollout_record_args()
    lib = _lib.load()
    for cls, fn in ((_lib.TaskStepArgs, "pulse_sizeof_task_step_args"), (_lib.TrajStepArgs, "pulse_sizeof_traj_step_args"),
                    (_lib.PdSimArgs, "pulse_sizeof_pd_sim_args"), (_lib.GemmX3pDesc, "pulse_sizeof_gemm_x3p_desc"),
                    (_lib.VaeEmbedArgs, "pulse_sizeof_vae_embed_args"), (_lib.AmpHistArgs, "pulse_sizeof_amp_hist_args"), (_lib.VaeKinArgs, "pulse_sizeof_vae_kin_args"),
                    (_lib.VaeHeadBwdArgs, "pulse_sizeof_vae_head_bwd_args"), (_lib.B16Transpose, "pulse_sizeof_b16_transpose")):
        assert ctypes.sizeof(cls) == getattr(lib, fn)(), fn
