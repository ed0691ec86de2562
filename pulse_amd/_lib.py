"""ctypes binding of libpulse_hip.so (the C ABI declared in include/pulse_hip.h).

The library is the product: there is NO CPU / PyTorch fallback.  If the shared
object is missing or cannot be loaded every op raises ``PulseLibraryError``
loudly (build it with ``python pulse_amd/csrc/build.py`` or
``__graft_entry__.build()``).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpulse_hip.so")
ABI_VERSION = 1

PULSE_IM_SELF_OBS = 1
PULSE_IM_TASK_OBS = 2
PULSE_IM_REWARD = 4
PULSE_IM_RESET = 8


class PulseLibraryError(RuntimeError):
    pass


class RewardSpecs(Structure):
    _fields_ = [("k_pos", c_float), ("k_rot", c_float), ("k_vel", c_float), ("k_ang_vel", c_float),
                ("w_pos", c_float), ("w_rot", c_float), ("w_vel", c_float), ("w_ang_vel", c_float),
                ("power_coef", c_float), ("power_reward", c_int32)]


class ImStepArgs(Structure):
    _fields_ = [
        ("rb", c_void_p), ("rb_env_stride", c_int64), ("num_envs", c_int32), ("num_bodies", c_int32),
        ("env_ids", c_void_p), ("num_ids", c_int32), ("env_mask", c_void_p),
        ("ref_now_pos", c_void_p), ("ref_now_rot", c_void_p), ("ref_now_vel", c_void_p), ("ref_now_ang", c_void_p),
        ("ref_next_pos", c_void_p), ("ref_next_rot", c_void_p), ("ref_next_vel", c_void_p), ("ref_next_ang", c_void_p),
        ("time_steps", c_int32),
        ("dof_force", c_void_p), ("dof_vel", c_void_p), ("num_dof", c_int32),
        ("progress", c_void_p), ("pass_time", c_void_p), ("cycle_counter", c_void_p),
        ("track_ids", c_void_p), ("num_track", c_int32), ("reset_ids", c_void_p), ("num_reset", c_int32),
        ("term_dist", c_void_p), ("reset_use_mean", c_int32), ("full_body_reward", c_int32),
        ("what", c_uint32), ("obs_version", c_int32), ("local_root_obs", c_int32), ("root_height_obs", c_int32),
        ("specs", RewardSpecs),
        ("obs", c_void_p), ("obs_stride", c_int64), ("obs_cols", c_int32),
        ("rew", c_void_p), ("rew_raw", c_void_p), ("reset", c_void_p), ("terminate", c_void_p),
    ]


P = c_void_p  # every device pointer crosses the ABI as void*

# name -> (restype, argtypes); must list EVERY symbol of include/pulse_hip.h
SIGNATURES = {
    "pulse_abi_version": (c_int, []),
    "pulse_last_error": (c_char_p, []),
    "pulse_quat_mul": (c_int, [P, P, P, c_int64, P]),
    "pulse_quat_conjugate": (c_int, [P, P, c_int64, P]),
    "pulse_quat_rotate": (c_int, [P, P, P, c_int64, P]),
    "pulse_quat_to_angle_axis": (c_int, [P, P, P, c_int64, P]),
    "pulse_quat_to_exp_map": (c_int, [P, P, c_int64, P]),
    "pulse_quat_to_tan_norm": (c_int, [P, P, c_int64, P]),
    "pulse_exp_map_to_quat": (c_int, [P, P, c_int64, P]),
    "pulse_slerp": (c_int, [P, P, P, P, c_int64, P]),
    "pulse_calc_heading": (c_int, [P, P, c_int64, P]),
    "pulse_calc_heading_quat": (c_int, [P, P, c_int64, c_int, P]),
    "pulse_sizeof_im_step_args": (c_int, []),
    "pulse_self_obs_width": (c_int, [c_int, c_int]),
    "pulse_task_obs_width": (c_int, [c_int, c_int, c_int]),
    "pulse_im_step": (c_int, [POINTER(ImStepArgs), P]),
    "pulse_gae": (c_int, [P, P, P, P, c_int32, c_int32, c_int64, c_int64, c_float, c_float, P, P, P]),
}

_lib = None
_load_error = None


def load():
    """Load (once) and return the ctypes library handle; raise loudly if unavailable."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise _load_error
    try:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} does not exist")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        v = lib.pulse_abi_version()
        if v != ABI_VERSION:
            raise OSError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    except (OSError, AttributeError) as exc:
        _load_error = PulseLibraryError(
            "pulse_amd: the HIP extension libpulse_hip.so is required and could not be loaded "
            f"({exc}). There is no CPU fallback. Build it with `python pulse_amd/csrc/build.py`.")
        raise _load_error from exc
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().pulse_last_error()
        raise PulseLibraryError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
