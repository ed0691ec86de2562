"""GPU: the action-dependent physics stand-in (pulse_pd_sim_step) against its CPU twin, and a short run of the return-parity
experiment (tools/return_parity.py; the 1000-iteration result is committed under profiles/)."""
import numpy as np
import pytest
import torch

from oracle import motion_oracle as MO
from pulse_amd import configs

pytestmark = pytest.mark.gpu


def test_pd_sim_lockstep_with_cpu_twin(dev):
    n, horizon, seed = 64, 16, 9
    env, _ = configs.make_env(n, horizon, dev, seed=seed, reference="motion_lib", env_overrides={"physics": "pd", "stateInit": "Start"})
    twin = MO.make_agent_env(n, horizon, seed, physics="pd", state_init_start=True)
    obs = env.reset()
    obs = obs["obs"] if isinstance(obs, dict) else obs
    o2 = twin.reset()
    np.testing.assert_allclose(obs.cpu().numpy(), o2.numpy(), atol=1e-5, rtol=1e-5)
    g = torch.Generator().manual_seed(1)
    for step in range(40):
        a = (0.8 * torch.randn(n, 69, generator=g)).clamp(-1.5, 1.5)
        obs, rew, done, info = env.step(a.to(dev))
        o2, r2, d2, i2 = twin.step(a)
        np.testing.assert_allclose(env.task.sim.err.cpu().numpy(), twin.inner.err.numpy(), atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(obs.cpu().numpy(), o2.numpy(), atol=3e-4, rtol=1e-4)
        np.testing.assert_allclose(rew.cpu().numpy(), r2.numpy(), atol=2e-5)
        assert torch.equal(done.cpu(), d2), step
        mask = done.bool()
        env.reset_masked(mask)
        twin.reset(torch.nonzero(d2).flatten())
    # the action matters: cancelling the sag gives a higher reward than doing nothing
    from pulse_amd import synthetic as syn
    sag, _ = syn.pd_sim_tables()
    rewards = {}
    for name, act in (("zero", torch.zeros(n, 69)), ("cancel", (-sag / syn.PD_SIM["action_scale"]).expand(n, -1).clamp(-1, 1))):
        env.reset()
        tot = 0.0
        for _ in range(12):
            _, rew, _, _ = env.step(act.to(dev))
            tot += rew.mean().item()
        rewards[name] = tot / 12
    assert rewards["cancel"] > rewards["zero"] + 0.05, rewards


def test_short_return_parity_run(dev):
    from tools.return_parity import run
    res = run(12, seed=3, device=str(dev), window=12, log=lambda *_: None)
    assert res["episodes_in_window"][0] > 0 and res["episodes_in_window"][0] == res["episodes_in_window"][1]
    assert res["relative_difference"] < 0.01 and res["relative_difference_step_reward"] < 0.01
