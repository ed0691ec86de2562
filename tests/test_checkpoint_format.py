"""CPU, build container only: a checkpoint written by the MI355X agent (tests/golden/ckpt_pulse_small.pth, generated on the
GPU by tools/make_ckpt_fixture.py) is parsed by the REFERENCE's own loaders (phc/learning/network_loader.py:76-176) and
driven through the reference's HumanoidZ.compute_z_actions source (phc/env/tasks/humanoid_z.py:81-155); the actions must
equal the ones the HIP decoder-in-env path produced from the same checkpoint."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import refload

FIX = os.path.join(os.path.dirname(__file__), "golden", "ckpt_pulse_small.pth")
pytestmark = pytest.mark.skipif(not (refload.available() and os.path.exists(FIX)), reason="needs the reference checkout and the GPU-made fixture")


def test_reference_loaders_read_our_checkpoint_and_reproduce_the_actions():
    ck = torch.load(FIX, map_location="cpu", weights_only=False)
    fx = ck["fixture"]
    f = refload.network_loader_functions()
    # key layout the reference's consumers look for
    keys = set(ck["model"].keys())
    for k in ("a2c_network.z_mlp.0.weight", "a2c_network.z_mu.weight", "a2c_network.z_logvar.weight", "a2c_network.actor_mlp.0.weight",
              "a2c_network.mu.weight", "a2c_network.z_prior.0.weight", "a2c_network.z_prior_mu.weight", "a2c_network.z_prior_logvar.weight",
              "a2c_network.critic_mlp.0.weight", "a2c_network.value.weight", "a2c_network.sigma"):
        assert k in keys, k
    assert set(ck["running_mean_std"].keys()) >= {"running_mean", "running_var", "count"}
    act_name = fx["network"]["mlp"]["activation"]
    enc = f["load_z_encoder"](ck, activation=act_name, z_type="vae", device="cpu")
    dec = f["load_z_decoder"](ck, activation=act_name, z_type="vae", device="cpu")
    units, tunits = fx["network"]["mlp"]["units"], fx["network"]["task_mlp"]["units"]
    assert [m.out_features for m in dec.decoder if isinstance(m, torch.nn.Linear)] == units + [69]
    assert [m.out_features for m in enc.encoder if isinstance(m, torch.nn.Linear)][:len(tunits)] == tunits
    assert dec.z_prior_mu.out_features == 32 and enc.z_mu.out_features == 32
    # the reference's decoder-in-env, from its own source, on the stored observations
    self_obs = 358
    stub = types.SimpleNamespace(
        get_self_obs_size=lambda: self_obs, obs_buf=fx["obs_buf"], running_mean=ck["running_mean_std"]["running_mean"],
        running_var=ck["running_mean_std"]["running_var"], distill_z_type="vae", use_vae_prior=True, use_vae_sphere_posterior=False,
        cfg={"env": {"embedding_norm": 1}}, z_all=False, decoder=dec, encoder=enc, is_discrete=False, embedding_size_distill=32)
    want = f["compute_z_actions"](stub, fx["action_z"].clone())
    np.testing.assert_allclose(fx["actions"].numpy(), want.numpy(), atol=3e-5, rtol=1e-4)
