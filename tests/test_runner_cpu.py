"""CPU: the rl_games plugin seam (phc/run_hydra.py:246-268) -- factories, builder names, config resolution -- without a GPU.
Where the reference tree is present (build container) the learner config dict is checked key-for-key against the reference's own
learning/im.yaml, so the config surface the runner accepts is the reference's."""
import os

import pytest

from pulse_amd import configs, runner as R

REF_YAML = "/root/reference/phc/data/cfg/learning/im.yaml"


def im_params():
    """learning/im.yaml as a dict (values from pulse_amd.configs, which cites the yaml line by line)."""
    cfg, _ = configs.agent_config("cfg2")
    conf = {k: v for k, v in cfg.items() if not k.startswith("_") and k != "network"}
    conf.update({"env_name": "rlgpu", "num_actors": 64, "save_frequency": 2500, "max_epochs": 10000000})
    return {"params": {"seed": 0, "algo": {"name": "im_amp"}, "model": {"name": "amp"}, "network": cfg["network"], "load_checkpoint": False,
                       "config": conf}}


def test_factories_and_names():
    r = R.build_alg_runner()
    for name in ("amp", "im_amp", "a2c_continuous"):
        assert name in r.algo_factory._builders
    for name in ("amp", "im_amp"):
        assert name in r.player_factory._builders
    for name in ("amp", "amp_z", "amp_z_reader"):
        assert name in r.model_builder.network_factory._builders
    with pytest.raises(ValueError):
        r.algo_factory.create("ppo_discrete")
    r.load(im_params())
    assert r.algo_name == "im_amp" and r.config["network"]["name"] == "amp" and r.config["seed"] == 0
    assert r.config["horizon_length"] == 32 and r.config["minibatch_size"] == 16384 and r.config["env_name"] == "rlgpu"


def test_env_registry():
    made = {}
    R.register_env("unit_test_env", lambda num_actors, **kw: made.setdefault("env", (num_actors, kw)))
    assert R.create_vec_env("unit_test_env", 7, a=1) == (7, {"a": 1})
    with pytest.raises(ValueError):
        R.create_vec_env("nope", 1)


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference tree not present")
def test_config_surface_matches_reference_yaml():
    import yaml
    with open(REF_YAML) as f:
        ref = yaml.safe_load(f)["params"]
    mine = im_params()["params"]
    assert mine["algo"]["name"] == ref["algo"]["name"] == "im_amp"
    assert mine["network"]["mlp"]["units"] == ref["network"]["mlp"]["units"] and mine["network"]["disc"]["units"] == ref["network"]["disc"]["units"]
    assert mine["network"]["space"]["continuous"]["sigma_init"]["val"] == ref["network"]["space"]["continuous"]["sigma_init"]["val"]
    skip = {"name", "score_to_win", "save_best_after", "print_stats", "save_intermediate", "player", "max_epochs", "save_frequency", "ppo"}
    for k, v in ref["config"].items():
        if k in skip:
            continue
        assert k in mine["config"], k
        if isinstance(v, (int, float, bool)):
            assert float(mine["config"][k]) == float(v), (k, mine["config"][k], v)


def test_dw_split_picks_the_fewest_slabs_that_fill_the_chip():
    """kernels.dw_split: host logic of the per-layer batch split of the weight-gradient GEMMs (no GPU needed)."""
    from pulse_amd.kernels import dw_split
    assert dw_split(128, 8) == 4          # cfg2 layer 1: 16 x 8 tiles -> 4 slabs = 512 workgroups
    assert dw_split(64, 8) == 8           # cfg2 layer 2 (2 nets x 4 x 8 tiles): all 8 slabs = 512 workgroups
    assert dw_split(4096, 8) == 1         # a huge layer needs no batch split at all
    assert dw_split(1, 8) == 8 and dw_split(1, 1) == 1
    assert dw_split(200, 6) == 3          # odd slab counts only halve while they stay even
    for tiles in (1, 7, 64, 100, 128, 500, 5000):
        for s in (1, 2, 4, 8, 16):
            d = dw_split(tiles, s)
            assert 1 <= d <= s and s % d == 0
            assert d == s or tiles * d >= 512                      # never below the fill target unless the maximum is
            assert d == 1 or d % 2 or tiles * (d // 2) < 512       # and never more slabs than needed
