"""GPU: an agent built the way phc/run_hydra.py builds it -- algo_factory.register_builder + Runner.load(im.yaml-shaped dict) +
Runner.run -- with the vec-env created from a registered creator (A2CBase's vecenv.create_vec_env); checkpoint -> player -> eval."""
import os

import numpy as np
import pytest
import torch

from pulse_amd import configs, runner as R
from tests.test_runner_cpu import im_params

pytestmark = pytest.mark.gpu


def _params(n=64):
    p = im_params()
    c = p["params"]["config"]
    c.update({"num_actors": n, "horizon_length": 16, "minibatch_size": 256, "amp_minibatch_size": 64, "amp_batch_size": 128,
              "amp_obs_demo_buffer_size": 4096, "amp_replay_buffer_size": 4096, "device": "cuda:0", "env_name": "pulse_motion_lib",
              "env_config": {"seed": 3}})
    p["params"]["network"]["mlp"]["units"] = [256, 128]
    p["params"]["network"]["disc"]["units"] = [256, 128]
    return p


def _creator(num_actors, seed=0, **kw):
    env, _ = configs.make_env(num_actors, 16, "cuda:0", seed=seed, env_kind="amp", reference="motion_lib")
    return env


def test_runner_builds_trains_saves_plays(dev, tmp_path):
    R.register_env("pulse_motion_lib", _creator)
    r = R.build_alg_runner()
    p = _params()
    p["params"]["config"]["train_dir"] = str(tmp_path)
    p["params"]["config"]["save_frequency"] = 2
    r.load(p)
    r.run({"train": True, "max_epochs": 2})
    ag = r.agent
    from pulse_amd.learning.im_amp import IMAmpAgent
    assert isinstance(ag, IMAmpAgent) and ag.enable_disc and ag.epoch_num == 2
    info = ag.vec_env.get_env_info()
    assert "amp_observation_space" in info and "enc_amp_observation_space" in info            # run_hydra.py:224-243
    ckpt = os.path.join(str(tmp_path), "nn", "Humanoid.pth")
    assert os.path.exists(ckpt)                                                                # save_frequency honoured
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert "a2c_network.actor_mlp.0.weight" in sd["model"] and "a2c_network._disc_mlp.0.weight" in sd["model"] and "amp_input_mean_std" in sd
    # evaluation (im_amp.py:136-363 reduced) and PMCP bookkeeping
    ev = ag.eval(max_steps=40)
    assert 0.0 <= ev["success_rate"] <= 1.0 and np.isfinite(ev["mpjpe_g"]) and ev["num_motions"] == 64
    ag.update_training_data(ev["failed_keys"])
    assert any(f.startswith("failed_") for f in os.listdir(ag.network_path))
    # the player registered under the same name restores the checkpoint and acts like the agent
    r2 = R.build_alg_runner()
    r2.load(p)
    out = r2.run({"train": False, "play": True, "checkpoint": ckpt, "n_steps": 20})
    assert "success_rate" in out
    pl = r2.player
    obs = pl.env.reset()
    a1 = pl.get_action(obs, is_determenistic=True)
    a2 = ag.get_action({"obs": ag.vec_env.reset()}, is_determenistic=True)
    assert a1.shape == a2.shape == (64, 69) and torch.isfinite(a1).all()
    for k, v in ag.model.state_dict().items():
        assert torch.equal(v, pl.model.state_dict()[k]), k
