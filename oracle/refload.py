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
        for part in methods_of.split("."):                   # "AMPZBuilder.Network": nested classes
            cls = [n for n in body if isinstance(n, ast.ClassDef) and n.name == part]
            assert cls, f"class {methods_of} not found in {path}"
            body = cls[0].body
    out = {}
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            text = "\n".join(lines[node.lineno - 1:node.end_lineno])  # excludes decorators
            if methods_of is not None:
                import textwrap
                text = textwrap.dedent(text)
                if text.lstrip().startswith("@"):           # never the case here (decorators are excluded), kept for safety
                    text = text[text.index("def "):]
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


HUMANOID_FUNCS = ["remove_base_rot", "compute_humanoid_observations_smpl_max", "compute_humanoid_observations_smpl_max_v2",
                  "compute_humanoid_observations_smpl_max_v3", "compute_humanoid_reset", "dof_to_obs_smpl"]
HUMANOID_IM_FUNCS = ["compute_imitation_observations", "compute_imitation_observations_v2", "compute_imitation_observations_v3",
                     "compute_imitation_observations_v6", "compute_imitation_observations_v7", "compute_imitation_observations_v8",
                     "compute_imitation_observations_v9", "compute_imitation_reward", "compute_humanoid_im_reset", "compute_point_goal_reward"]
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


AMP_AGENT_METHODS = ["_disc_loss", "_disc_loss_neg", "_disc_loss_pos", "_compute_disc_acc", "_calc_disc_rewards", "_calc_amp_rewards",
                     "_eval_disc", "_preproc_amp_obs", "_combine_rewards", "_optimize_kin", "_assamble_kin_dict", "_norm_disc_reward",
                     "calc_gradients", "_preproc_obs", "play_steps"]
AMP_NET_METHODS = ["eval_disc", "get_disc_logit_weights", "get_disc_weights"]
AMPZ_NET_METHODS = ["form_embedding", "compute_prior", "reparameterize", "eval_actor", "eval_critic"]
AMPSEPT_NET_METHODS = ["eval_task", "eval_actor", "eval_critic"]


def learning_methods():
    """Methods of AMPAgent (phc/learning/amp_agent.py), AMPBuilder.Network (amp_network_builder.py) and AMPZBuilder.Network
    (amp_network_z_builder.py) extracted by name: the modules import rl_games at top level, the method BODIES are plain torch.
    Returns {'agent': {...}, 'amp_net': {...}, 'ampz_net': {...}} of functions taking an explicit ``self``."""
    if "learning" in _cache:
        return _cache["learning"]
    import torch
    ns = _namespace()
    tu = torch_utils()
    lf = importable_modules()["loss_functions"]
    ns.update({"kl_multi": lf.kl_multi, "project_to_norm": tu.project_to_norm, "to_torch": ns.get("to_torch"),
               "flags": types.SimpleNamespace(test=False, trigger_input=False, debug=False), "F": torch.nn.functional})
    from . import agent_oracle as _ao                        # torch_ext.policy_kl is rl_games (absent): the restated form, SURVEY Appendix B
    ns["torch_ext"] = types.SimpleNamespace(policy_kl=_ao.policy_kl)
    ns["a2c_common"] = types.SimpleNamespace(swap_and_flatten01=_ao.swap_and_flatten01)   # rl_games a2c_common.swap_and_flatten01 (Appendix B)
    base = os.path.join(REFERENCE_ROOT, "phc", "learning")
    out = {}
    for key, fname, cls, names in (("agent", "amp_agent.py", "AMPAgent", AMP_AGENT_METHODS),
                                   ("amp_net", "amp_network_builder.py", "AMPBuilder.Network", AMP_NET_METHODS),
                                   ("ampz_net", "amp_network_z_builder.py", "AMPZBuilder.Network", AMPZ_NET_METHODS),
                                   ("ampsept_net", "amp_network_sept_builder.py", "AMPSeptBuilder.Network", AMPSEPT_NET_METHODS)):
        srcs = _extract(os.path.join(base, fname), names, methods_of=cls)
        if key == "agent":                                   # inherited from CommonAgent
            srcs.update(_extract(os.path.join(base, "common_agent.py"), ["_actor_loss", "_critic_loss", "bound_loss", "get_action_values", "_eval_critic",
                                                                         "discount_values"], methods_of="CommonAgent"))
        local = dict(ns)
        for name, text in srcs.items():
            exec(compile(text, f"<reference:{cls}.{name}>", "exec"), local)
        out[key] = {k: local[k] for k in srcs}
    _cache["learning"] = out
    return out


HUMANOID_IM_METHODS = ["_compute_task_obs", "_compute_reward", "_compute_reset", "_get_state_from_motionlib_cache", "_action_to_pd_targets"]


def humanoid_im_methods():
    """Step-composition methods of HumanoidIm (phc/env/tasks/humanoid_im.py:708-919, 950-964, 1119-1192) extracted by name; they call
    the jit functions of env_functions() and a motion library with MotionLibBase's query surface."""
    if "him" in _cache:
        return _cache["him"]
    ns = _namespace()
    ns.update(env_functions())
    ns["flags"] = types.SimpleNamespace(test=False, im_eval=False, no_collision_check=False, real_traj=False)   # run_hydra.py defaults
    path = os.path.join(REFERENCE_ROOT, "phc", "env", "tasks", "humanoid_im.py")
    srcs = _extract(path, HUMANOID_IM_METHODS, methods_of="HumanoidIm")
    for name, text in srcs.items():
        exec(compile(text, f"<reference:HumanoidIm.{name}>", "exec"), ns)
    _cache["him"] = {k: ns[k] for k in srcs}
    return _cache["him"]


def pnn_reference():
    """load_pnn / load_mcp_mlp (phc/learning/network_loader.py:11-73) and the PNN class body (phc/learning/pnn.py:9-131), rebuilt on
    nn.Module with the rl_games activation factory stubbed by torch_utils.activation_facotry."""
    if "pnn" in _cache:
        return _cache["pnn"]
    import collections
    import torch
    ns = _namespace()
    tu = torch_utils()
    ns.update({"defaultdict": collections.defaultdict, "edict": None})
    methods = _extract(os.path.join(REFERENCE_ROOT, "phc", "learning", "pnn.py"), ["__init__", "freeze_pnn", "_build_sequential_mlp", "forward"], methods_of="PNN")
    body = {}
    for name, text in methods.items():
        exec(compile(text, f"<reference:PNN.{name}>", "exec"), ns, body)
    body["activations_factory"] = types.SimpleNamespace(create=lambda name: tu.activation_facotry(name)())
    ns["PNN"] = type("PNN", (torch.nn.Module,), body)
    srcs = _extract(os.path.join(REFERENCE_ROOT, "phc", "learning", "network_loader.py"), ["load_mcp_mlp", "load_pnn"])
    for name, text in srcs.items():
        exec(compile(text, f"<reference:{name}>", "exec"), ns)
    _cache["pnn"] = {"PNN": ns["PNN"], "load_pnn": ns["load_pnn"], "load_mcp_mlp": ns["load_mcp_mlp"]}
    return _cache["pnn"]


HUMANOID_AMP_METHODS = ["_update_hist_amp_obs", "_compute_amp_observations", "_compute_amp_observations_from_state", "_init_amp_obs",
                        "_init_amp_obs_default", "_init_amp_obs_ref", "_get_state_from_motionlib_cache"]


def humanoid_amp_methods():
    """AMP-window methods of HumanoidAMP (phc/env/tasks/humanoid_amp.py:519-563, 622-680) extracted by name."""
    if "hamp" in _cache:
        return _cache["hamp"]
    ns = _namespace()
    ns.update(env_functions())
    ns["flags"] = types.SimpleNamespace(test=False, real_traj=False)
    srcs = _extract(os.path.join(REFERENCE_ROOT, "phc", "env", "tasks", "humanoid_amp.py"), HUMANOID_AMP_METHODS, methods_of="HumanoidAMP")
    for name, text in srcs.items():
        exec(compile(text, f"<reference:HumanoidAMP.{name}>", "exec"), ns)
    _cache["hamp"] = {k: ns[k] for k in srcs}
    return _cache["hamp"]


def task_functions():
    """The downstream-task TorchScript functions (README's PULSE commands): compute_humanoid_reset (humanoid.py:1572-1608),
    compute_speed_observations / _reward (humanoid_speed.py:310-343), compute_location_observations / compute_reach_reward
    (humanoid_reach.py:224-250), compute_strike_observations / _reward / its compute_humanoid_reset (humanoid_strike.py:270-380)."""
    if "tasks" in _cache:
        return _cache["tasks"]
    tasks = os.path.join(REFERENCE_ROOT, "phc", "env", "tasks")
    out = {}
    for fname, names, prefix in (("humanoid.py", ["compute_humanoid_reset"], ""),
                                 ("humanoid_speed.py", ["compute_speed_observations", "compute_speed_reward"], ""),
                                 ("humanoid_reach.py", ["compute_location_observations", "compute_reach_reward"], ""),
                                 ("humanoid_strike.py", ["compute_strike_observations", "compute_strike_reward", "compute_humanoid_reset"], "strike_")):
        ns = _namespace()
        for name, text in _extract(os.path.join(tasks, fname), names).items():
            exec(compile(text, f"<reference:{fname}:{name}>", "exec"), ns)
            out[(prefix if name == "compute_humanoid_reset" else "") + name] = ns[name]
    _cache["tasks"] = out
    return out


def task_reward_methods():
    """HumanoidSpeed._compute_reward (humanoid_speed.py:198-240) and HumanoidStrike._compute_reward (humanoid_strike.py:174-199) extracted by
    name, with their jit functions in scope: the power_reward / power_usage_reward terms are written in the method bodies, not in the jit
    functions."""
    if "task_rew" in _cache:
        return _cache["task_rew"]
    tasks = os.path.join(REFERENCE_ROOT, "phc", "env", "tasks")
    tf = task_functions()
    out = {}
    for fname, cls in (("humanoid_speed.py", "HumanoidSpeed"), ("humanoid_strike.py", "HumanoidStrike")):
        ns = _namespace()
        ns.update({k: tf[k] for k in ("compute_speed_reward", "compute_strike_reward")})
        ns["flags"] = types.SimpleNamespace(test=False)
        for name, text in _extract(os.path.join(tasks, fname), ["_compute_reward"], methods_of=cls).items():
            exec(compile(text, f"<reference:{cls}.{name}>", "exec"), ns)
            out[cls] = ns[name]
    _cache["task_rew"] = out
    return out


def terrain_functions():
    """HumanoidTraj / HumanoidPedestrianTerrain (README: the terrain-traversal PULSE command): the TorchScript functions
    compute_location_observations (with the upright flag), compute_location_reward(_fuzzy), quat_apply_yaw and the terrain variant of
    compute_humanoid_reset (humanoid_pedestrian_terrain.py:1476-1646), HumanoidTraj's compute_humanoid_reset (humanoid_traj.py:256-300),
    the METHODS get_heights / get_center_heights (:690-772), _fetch_traj_samples (humanoid_traj.py:196-211) and Terrain.world_points_to_map /
    sample_height_points (:1191-1270), plus the TrajGenerator class (phc/utils/traj_generator.py; its ``np.int`` needs numpy < 1.24, so
    the name is supplied)."""
    if "terrain" in _cache:
        return _cache["terrain"]
    import numpy as np
    tasks = os.path.join(REFERENCE_ROOT, "phc", "env", "tasks")
    out = {}
    ns = _namespace()
    flags = types.SimpleNamespace(divide_group=False, no_collision_check=False, fixed_path=False, slow=False, real_path=False, height_debug=False)
    ns["flags"] = flags
    # HumanoidTraj's own jit functions first (the terrain file re-defines two of the names)
    for name, text in _extract(os.path.join(tasks, "humanoid_traj.py"), ["compute_humanoid_reset", "compute_location_observations"]).items():
        ns_t = dict(ns)
        exec(compile(text, f"<reference:humanoid_traj.py:{name}>", "exec"), ns_t)
        out["traj_" + name] = ns_t[name]
    tf = os.path.join(tasks, "humanoid_pedestrian_terrain.py")
    for name, text in _extract(tf, ["compute_humanoid_reset", "quat_apply_yaw", "compute_location_observations", "compute_location_reward",
                                    "compute_location_reward_fuzzy"]).items():
        exec(compile(text, f"<reference:humanoid_pedestrian_terrain.py:{name}>", "exec"), ns)
        out["terrain_" + name if name == "compute_humanoid_reset" else name] = ns[name]
    ns["remove_base_rot"] = env_functions()["remove_base_rot"]
    for cls, names in (("HumanoidPedestrianTerrain", ["get_heights", "get_center_heights"]), ("Terrain", ["world_points_to_map", "sample_height_points"])):
        for name, text in _extract(tf, names, methods_of=cls).items():
            exec(compile(text, f"<reference:{cls}.{name}>", "exec"), ns)
            out[f"{cls}.{name}"] = ns[name]
    for name, text in _extract(os.path.join(tasks, "humanoid_traj.py"), ["_fetch_traj_samples"], methods_of="HumanoidTraj").items():
        exec(compile(text, f"<reference:HumanoidTraj.{name}>", "exec"), ns)
        out[f"HumanoidTraj.{name}"] = ns[name]
    if not hasattr(np, "int"):
        np.int = int                                                 # traj_generator.py:54 (removed from numpy 1.24)
    _ensure_paths()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_traj_generator", os.path.join(REFERENCE_ROOT, "phc", "utils", "traj_generator.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault("joblib", __import__("joblib"))
    spec.loader.exec_module(mod)
    out["TrajGenerator"] = mod.TrajGenerator
    for k in ("fixed_path", "slow", "real_path"):                    # run_hydra.py:284-301 sets these on the global flags object; training: all False
        if not hasattr(mod.flags, k):
            setattr(mod.flags, k, False)
    out["flags"] = mod.flags
    _cache["terrain"] = out
    return out


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
