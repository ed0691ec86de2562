"""Load the reference's own hot-path functions from /root/reference.

ORACLE / TEST INFRASTRUCTURE -- used only by ``oracle/gen_golden.py`` (and the
optional ``-m "not gpu"`` test that re-derives the goldens when the reference
tree is present).  /root/reference does not exist on the GPU box; nothing that
runs there imports this module.

Mechanism:
  * ``oracle/shim`` supplies a file-backed ``isaacgym.torch_utils``;
  * ``phc/utils/torch_utils.py`` is then imported UNMODIFIED;
  * the ``@torch.jit.script`` functions of ``phc/env/tasks/humanoid*.py`` cannot
    be imported (their modules pull in isaacgym.gymapi, smpl_sim, open3d ...),
    so they are AST-extracted by name, the decorator is dropped and the source
    is exec'd eagerly against the namespace the original module would have had;
  * a few pure methods of ``CommonAgent`` (GAE, PPO losses) are extracted the
    same way and called with a stub ``self``.
No reference source is copied into this repository: the text is read from
/root/reference at run time.
"""
import ast
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PULSE_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "phc", "utils", "torch_utils.py"))


def _ensure_paths():
    for p in (_SHIM, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)


_cache = {}


def torch_utils():
    """The reference's phc.utils.torch_utils module, imported unmodified."""
    if "tu" not in _cache:
        _ensure_paths()
        _cache["tu"] = importlib.import_module("phc.utils.torch_utils")
    return _cache["tu"]


def _extract(path, names, *, methods_of=None):
    """Return {name: source} for top-level functions (or methods of a class)."""
    with open(path, "r") as f:
        src = f.read()
    tree = ast.parse(src)
    lines = src.splitlines()
    body = tree.body
    if methods_of is not None:
        cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == methods_of]
        assert cls, f"class {methods_of} not found in {path}"
        body = cls[0].body
    out = {}
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            text = "\n".join(lines[node.lineno - 1:node.end_lineno])  # excludes decorators
            if methods_of is not None:
                import textwrap
                text = textwrap.dedent(text)
            out[node.name] = text
    missing = set(names) - set(out)
    assert not missing, f"not found in {path}: {missing}"
    return out


def _namespace():
    import numpy as np
    import torch
    tu = torch_utils()
    ns = {"torch": torch, "np": np, "torch_utils": tu, "nn": torch.nn}
    iso = importlib.import_module("isaacgym.torch_utils")
    for k in dir(iso):
        if not k.startswith("_"):
            ns[k] = getattr(iso, k)
    from typing import Dict, List, Tuple
    ns.update({"Dict": Dict, "List": List, "Tuple": Tuple})
    return ns


HUMANOID_FUNCS = ["remove_base_rot", "compute_humanoid_observations_smpl_max", "compute_humanoid_reset",
                  "dof_to_obs_smpl"]
HUMANOID_IM_FUNCS = ["compute_imitation_observations_v6", "compute_imitation_observations_v7",
                     "compute_imitation_reward", "compute_humanoid_im_reset"]
HUMANOID_AMP_FUNCS = ["build_amp_observations_smpl"]
COMMON_AGENT_METHODS = ["discount_values", "_actor_loss", "_critic_loss", "bound_loss", "_calc_advs"]


def env_functions():
    """dict name -> callable, the reference's jit functions executed eagerly."""
    if "env" in _cache:
        return _cache["env"]
    ns = _namespace()
    tasks = os.path.join(REFERENCE_ROOT, "phc", "env", "tasks")
    srcs = {}
    srcs.update(_extract(os.path.join(tasks, "humanoid.py"), HUMANOID_FUNCS))
    srcs.update(_extract(os.path.join(tasks, "humanoid_im.py"), HUMANOID_IM_FUNCS))
    srcs.update(_extract(os.path.join(tasks, "humanoid_amp.py"), HUMANOID_AMP_FUNCS))
    # order matters only for call-time lookups, all land in the same namespace
    for name, text in srcs.items():
        exec(compile(text, f"<reference:{name}>", "exec"), ns)
    _cache["env"] = {k: ns[k] for k in srcs}
    return _cache["env"]


def agent_methods():
    """dict name -> function(self, ...) extracted from CommonAgent."""
    if "agent" in _cache:
        return _cache["agent"]
    ns = _namespace()
    path = os.path.join(REFERENCE_ROOT, "phc", "learning", "common_agent.py")
    srcs = _extract(path, COMMON_AGENT_METHODS, methods_of="CommonAgent")
    for name, text in srcs.items():
        exec(compile(text, f"<reference:CommonAgent.{name}>", "exec"), ns)
    _cache["agent"] = {k: ns[k] for k in srcs}
    return _cache["agent"]


MOTION_LIB_METHODS = ["get_motion_state", "get_root_pos_smpl", "_calc_frame_blend", "_get_num_bodies", "_local_rotation_to_dof_smpl",
                      "get_motion_num_steps", "get_motion_length", "sample_time", "sample_time_interval"]


def motion_lib_class():
    """A class carrying MotionLibBase's query methods (phc/utils/motion_lib_base.py:396-564), extracted by name and
    exec'd against the namespace the module would have had (``flags.real_traj`` False, as in every training config).
    The module itself cannot be imported (smpl_sim, joblib-loaded AMASS pickles, multiprocessing loaders)."""
    if "motion_lib" in _cache:
        return _cache["motion_lib"]
    ns = _namespace()
    ns["flags"] = types.SimpleNamespace(real_traj=False)
    path = os.path.join(REFERENCE_ROOT, "phc", "utils", "motion_lib_base.py")
    srcs = _extract(path, MOTION_LIB_METHODS, methods_of="MotionLibBase")
    for name, text in srcs.items():
        exec(compile(text, f"<reference:MotionLibBase.{name}>", "exec"), ns)
    cls = type("ReferenceMotionLibQueries", (), {k: ns[k] for k in srcs})
    _cache["motion_lib"] = cls
    return cls


class _AttrDict(dict):
    """easydict.EasyDict stand-in (easydict is not installed): attribute access on a dict."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def network_loader_functions():
    """load_mlp / load_linear / load_z_encoder / load_z_decoder of phc/learning/network_loader.py:76-176 (the module imports
    easydict and the PNN / VQ classes at top level, so the functions are extracted by name) and HumanoidZ.compute_z_actions
    (phc/env/tasks/humanoid_z.py:81-155)."""
    if "netload" in _cache:
        return _cache["netload"]
    ns = _namespace()
    tu = torch_utils()
    ns.update({"edict": _AttrDict, "project_to_norm": tu.project_to_norm, "F": __import__("torch").nn.functional})
    srcs = _extract(os.path.join(REFERENCE_ROOT, "phc", "learning", "network_loader.py"), ["load_mlp", "load_linear", "load_z_encoder", "load_z_decoder"])
    srcs.update(_extract(os.path.join(REFERENCE_ROOT, "phc", "env", "tasks", "humanoid_z.py"), ["compute_z_actions"], methods_of="HumanoidZ"))
    for name, text in srcs.items():
        exec(compile(text, f"<reference:{name}>", "exec"), ns)
    _cache["netload"] = {k: ns[k] for k in srcs}
    return _cache["netload"]


def importable_modules():
    """Reference modules that import cleanly here (no shim needed)."""
    _ensure_paths()
    return {
        "running_mean_std": importlib.import_module("phc.utils.running_mean_std"),
        "replay_buffer": importlib.import_module("phc.learning.replay_buffer"),
        "loss_functions": importlib.import_module("phc.learning.loss_functions"),
        "rotation3d": importlib.import_module("poselib.poselib.core.rotation3d"),
    }


def stub_self(**kw):
    return types.SimpleNamespace(**kw)
