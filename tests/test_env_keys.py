"""CPU: no env option the reference reads on this path can be dropped silently (pulse_amd/env/env_keys.py).

  * when /root/reference is mounted: the set of ``cfg["env"]`` keys read by the reference's task classes is re-derived from its sources
    and every one of them must be classified (HONOURED / INERT / UNBUILT);
  * every key classified HONOURED is actually read somewhere in pulse_amd/env or pulse_amd/learning;
  * every shipped phc/data/cfg/env/*.yaml either passes the audit or raises NotImplementedError naming the offending key -- the expected
    outcome per file is written down below;
  * switching any UNBUILT option on raises by name.
(The GPU half -- the envs really construct from these dicts -- is tests/test_zero_out_far_gpu.py / test_env_keys_gpu.py.)"""
import glob
import os
import re
import warnings

import pytest
import yaml

from oracle import refload
from pulse_amd.env import env_keys as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TASK_FILES = ["humanoid.py", "humanoid_amp.py", "humanoid_amp_task.py", "humanoid_im.py", "humanoid_z.py", "humanoid_im_getup.py", "humanoid_im_distill.py",
              "humanoid_im_distill_getup.py", "base_task.py", "humanoid_speed.py", "humanoid_reach.py", "humanoid_strike.py", "humanoid_traj.py",
              "humanoid_pedestrian_terrain.py", "vec_task.py", "vec_task_wrappers.py"]
needs_reference = pytest.mark.skipif(not refload.available(), reason="reference checkout not mounted")


def _reference_keys():
    pat = re.compile(r"""cfg\[['"]env['"]\](?:\.get\(|\[)['"]([A-Za-z_0-9]+)['"]""")
    keys = set()
    for f in TASK_FILES:
        with open(os.path.join(refload.REFERENCE_ROOT, "phc", "env", "tasks", f)) as fh:
            keys.update(pat.findall(fh.read()))
    for f in glob.glob(os.path.join(refload.REFERENCE_ROOT, "phc", "learning", "*.py")):
        with open(f) as fh:
            src = fh.read()
        keys.update(pat.findall(src))
        keys.update(re.findall(r"""task\.cfg\.env\.get\(['"]([A-Za-z_0-9]+)['"]""", src))
    return keys


@needs_reference
def test_every_key_the_reference_reads_is_classified():
    keys = _reference_keys()
    assert len(keys) > 120, "the source scan found implausibly few keys"
    missing = sorted(keys - K.ALL_KNOWN)
    assert not missing, f"env keys read by the reference but not classified in env_keys.py: {missing}"
    # the three classes are disjoint
    assert not (K.HONOURED & set(K.INERT)) and not (K.HONOURED & set(K.UNBUILT)) and not (set(K.INERT) & set(K.UNBUILT))


def test_honoured_keys_are_really_read():
    src = ""
    for d in ("env", "learning"):
        for f in glob.glob(os.path.join(ROOT, "pulse_amd", d, "*.py")):
            if not f.endswith("env_keys.py"):
                with open(f) as fh:
                    src += fh.read()
    with open(os.path.join(ROOT, "pulse_amd", "configs.py")) as fh:
        src += fh.read()
    unread = sorted(k for k in K.HONOURED if not re.search(r"""['"]%s['"]""" % re.escape(k), src))
    assert not unread, f"classified HONOURED but never read by pulse_amd: {unread}"


# file -> None (passes the audit) or the key whose shipped value is not built
EXPECTED = {
    "env_im.yaml": None, "env_im_pnn.yaml": None, "env_vr.yaml": None, "env_im_vae.yaml": None, "env_im_getup_mcp.yaml": None,
    "env_pulse_amp.yaml": None, "env_pulse_im.yaml": None, "env_pulsex_amp.yaml": None, "env_pulse_terrain.yaml": None,
    "phc_kp_mcp_iccv.yaml": None, "phc_kp_pnn_iccv.yaml": None, "phc_prim_iccv.yaml": None, "phc_prim_vr.yaml": None,
    "phc_shape_mcp_iccv.yaml": None, "phc_shape_pnn_iccv.yaml": None, "phc_shape_pnn_train_iccv.yaml": None,
}


@needs_reference
def test_yaml_walk_every_shipped_env_config_constructs_or_raises_by_name():
    files = sorted(glob.glob(os.path.join(refload.REFERENCE_ROOT, "phc", "data", "cfg", "env", "*.yaml")))
    assert {os.path.basename(f) for f in files} == set(EXPECTED), "a shipped env config is not covered by the walk"
    zof = []
    for f in files:
        with open(f) as fh:
            d = yaml.safe_load(fh)
        env = d["env"] if isinstance(d.get("env"), dict) else d                # legacy files nest their options under env:
        name = os.path.basename(f)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                                   # (unread legacy spellings are reported as warnings)
            if EXPECTED[name] is None:
                K.audit(env, name)
            else:
                with pytest.raises(NotImplementedError, match=EXPECTED[name]):
                    K.audit(env, name)
        if env.get("zero_out_far"):
            zof.append(name)
    # the five configs round 5 silently mis-computed
    assert zof == ["env_im_getup_mcp.yaml", "phc_kp_mcp_iccv.yaml", "phc_kp_pnn_iccv.yaml", "phc_shape_mcp_iccv.yaml", "phc_shape_pnn_iccv.yaml"]


@pytest.mark.parametrize("key", sorted(K.UNBUILT))
def test_unbuilt_options_raise_by_name(key):
    ok, _ = K.UNBUILT[key]
    bad = {"numAMPEncObsSteps": 4, "control_mode": "pd", "amp_obs_v": 2}.get(key, True)
    with pytest.raises(NotImplementedError, match=key):
        K.audit({key: bad, "numAMPObsSteps": 10}, "test")
    good = 10 if ok == "==numAMPObsSteps" else ok[0]
    K.audit({key: good, "numAMPObsSteps": 10}, "test")


def test_unknown_keys_warn_once():
    K._warned.discard("definitely_not_a_key")
    with pytest.warns(UserWarning, match="definitely_not_a_key"):
        K.audit({"definitely_not_a_key": 1}, "test")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        K.audit({"definitely_not_a_key": 1}, "test")
