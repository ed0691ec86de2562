"""GPU: one full train_epoch of the HIP-backed CommonAgent vs the PyTorch-CPU oracle agent on the
SAME recorded rollout, initial weights, sampling noise and minibatch permutations.

north_star bars: bit-exact env / body indexing (integer outputs, row order), fp32 rewards and
advantages within 1e-5, policy grad-norm within 1e-4 (relative).
"""
import copy

import numpy as np
import pytest
import torch

from oracle import agent_oracle as AO
from pulse_amd import configs, synthetic as syn
from pulse_amd.env.sim import RecordedRollout

pytestmark = pytest.mark.gpu


def build_pair(name, dev, seed=7, epochs=1, **overrides):
    cfg, num_envs = configs.agent_config(name, **overrides)
    t = cfg["horizon_length"]
    rollout_cpu = RecordedRollout(num_envs, t + 1, seed=seed)
    rollout_dev = copy.deepcopy(rollout_cpu)
    noise = torch.randn(epochs, t, num_envs, 69, generator=torch.Generator().manual_seed(seed))
    torch.manual_seed(seed)
    oenv = AO.OracleEnv(rollout_cpu, syn.RESET_BODY_IDS, list(range(24)))
    oracle = AO.OracleCommonAgent(cfg, oenv, cfg["network"]["mlp"]["units"], seed=seed, noise=noise)
    # permutation_device="cpu": minibatches drawn on the host like the reference, so the shared seed reproduces the oracle's batches
    agent, _ = configs.make_agent(name, device=str(dev), seed=seed, rollout=rollout_dev, permutation_device="cpu", **overrides)
    agent.model.load_state_dict(oracle.model.state_dict_ref())
    noise_dev = noise.to(dev)
    agent.noise_provider = lambda e, s: noise_dev[e, s]
    return oracle, agent


def cmp(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=what)


@pytest.mark.parametrize("name,over", [("cfg1", {}), ("cfg1", {"clip_value": True, "minibatch_size": 512})])
def test_train_epoch_parity(dev, name, over):
    oracle, agent = build_pair(name, dev, **over)
    ref = oracle.train_epoch()
    info = agent.train_epoch()
    eb = agent.experience_buffer
    td, rd = eb.tensor_dict, oracle.tensor_dict
    # ---- rollout: env side
    cmp(td["obses"], rd["obses"], 1e-5, 1e-5, "obses")
    cmp(td["next_obses"], rd["next_obses"], 1e-5, 1e-5, "next_obses")
    cmp(td["rewards"], rd["rewards"], 1e-5, what="rewards")                       # fp32 rewards within 1e-5
    assert torch.equal(td["dones"].cpu(), rd["dones"]), "dones must be bit-exact"
    assert rd["dones"].sum() > 0
    # ---- rollout: policy side (fp32 GEMMs, different summation order than MKL)
    cmp(td["mus"], rd["mus"], 2e-5, 1e-5, "mus")
    cmp(td["actions"], rd["actions"], 2e-5, 1e-5, "actions")
    cmp(td["values"], rd["values"], 2e-5, 1e-5, "values")
    cmp(td["next_values"], rd["next_values"], 2e-5, 1e-5, "next_values")
    cmp(td["neglogpacs"], rd["neglogpacs"], 2e-3, 1e-5, "neglogpacs")             # ~|a-mu|/sigma^2 * 2e-5 amplification
    # ---- GAE: fp32 advantages within 1e-5 of the oracle run on the SAME stored inputs ...
    from oracle import env_oracle as E
    adv_same = E.gae(td["dones"].float().cpu(), td["values"].cpu(), td["rewards"].cpu(), td["next_values"].cpu(), 0.99, 0.95)
    adv_dev = agent.discount_values(td["dones"], td["values"], td["rewards"], td["next_values"])
    cmp(adv_dev, adv_same, 1e-5, what="advantages (same inputs)")
    # ... and end to end (value-head round-off propagates through the discounted sum)
    cmp(info_batch(agent)["advs_raw"], ref["batch_dict"]["advs_raw"], 2e-4, 1e-5, "advantages (end to end)")
    # flatten order is env-major in both (row = env * T + t)
    cmp(eb.flat("obses")[:, :934], ref["batch_dict"]["obses"], 1e-5, 1e-5, "flattened obses")
    # ---- dataset
    ds = agent.dataset.values_dict
    cmp(ds["advantages"], oracle.values_dict["advantages"], 5e-4, 1e-4, "normalised advantages")
    cmp(ds["returns"], oracle.values_dict["returns"], 2e-4, 1e-4, "normalised returns")
    cmp(agent.value_mean_std.running_mean, oracle.value_mean_std.running_mean, 1e-5, 1e-5)
    # ---- update: per-minibatch losses and the gradient norm
    n_mb = len(ref["infos"])
    assert n_mb == len(info["actor_loss"]) == agent.mini_epochs_num * agent.num_minibatches
    gn_dev = torch.stack(info["grad_norm"]).reshape(-1).cpu().double().numpy()
    gn_ref = np.array(oracle.grad_norms)
    np.testing.assert_allclose(gn_dev[0], gn_ref[0], rtol=1e-4)                   # policy grad-norm within 1e-4
    np.testing.assert_allclose(gn_dev, gn_ref, rtol=2e-3)                         # later steps inherit weight round-off
    for key, okey in (("actor_loss", "actor_loss"), ("critic_loss", "critic_loss"), ("b_loss", "b_loss"), ("kl", "kl")):
        dev_v = torch.stack(info[key]).cpu().double().numpy()
        ref_v = np.array([float(x[okey]) for x in ref["infos"]])
        np.testing.assert_allclose(dev_v, ref_v, rtol=2e-3, atol=2e-4, err_msg=key)
    # ---- final state
    # Adam's g / sqrt(v) is +-1 for |g| ~ round-off, so a handful of weights whose gradient is ~0 move by up to
    # lr per step in either direction: bound those by steps * 2 lr and require the bulk to agree tightly.
    sd = agent.model.state_dict()
    bound = 2.0 * agent.last_lr * n_mb
    for k, v in oracle.model.state_dict_ref().items():
        d = (sd[k].cpu().double() - v.double()).abs()
        assert d.max().item() <= bound, k
        assert (d > 2e-6 + 1e-4 * v.double().abs()).double().mean().item() < 0.03, k
    cmp(agent.running_mean_std.running_mean, oracle.running_mean_std.running_mean, 1e-5, 1e-5)
    cmp(agent.running_mean_std.running_var, oracle.running_mean_std.running_var, 1e-5, 1e-4)
    assert agent.running_mean_std.count.item() == oracle.running_mean_std.count.item()


def info_batch(agent):
    eb = agent.experience_buffer
    td = eb.tensor_dict
    adv = agent.discount_values(td["dones"], td["values"], td["rewards"], td["next_values"])
    return {"advs_raw": adv.transpose(0, 1).reshape(-1, 1)}


def test_second_epoch_stays_in_step(dev):
    """Two epochs back to back (stats carried over, envs reset across the epoch seam)."""
    oracle, agent = build_pair("cfg1", dev, seed=11, epochs=2)
    for _ in range(2):
        ref = oracle.train_epoch()
        info = agent.train_epoch()
    td, rd = agent.experience_buffer.tensor_dict, oracle.tensor_dict
    cmp(td["rewards"], rd["rewards"], 1e-5)
    assert torch.equal(td["dones"].cpu(), rd["dones"])
    cmp(td["obses"], rd["obses"], 1e-5, 1e-5)
    cmp(td["values"], rd["values"], 5e-4, 1e-3)
    gn_dev = torch.stack(info["grad_norm"]).reshape(-1).cpu().double().numpy()
    np.testing.assert_allclose(gn_dev[0], oracle.grad_norms[len(oracle.grad_norms) // 2], rtol=2e-3)


def test_reference_style_gathered_minibatch(dev):
    """calc_gradients also accepts the reference's gathered input_dict (AMPDataset._get_item)."""
    _, agent = build_pair("cfg1", dev, seed=5)
    agent.train_epoch()
    d = agent.dataset.gather(0)
    before = agent.model.flat.clone()
    agent.calc_gradients({k: d[k] for k in ("old_values", "old_logp_actions", "advantages", "returns", "actions", "obs", "mu", "sigma")})
    assert torch.isfinite(agent.train_result["actor_loss"]).item()
    assert not torch.equal(before, agent.model.flat)


def test_bootstrap_value_shortcut_equals_full_critic_pass(dev):
    """CommonAgent._bootstrap_values reuses values[t + 1] as the bootstrap value of step t for envs that were not reset: the
    result must equal a critic pass over every next observation (amp_agent.py:394-398), and the shortcut must switch itself off for an
    env wrapper that does not declare ``obs_carries_over``."""
    def played(carries):
        torch.manual_seed(99)                                             # identical initial weights and rollouts
        agent, _ = configs.make_agent("cfg1", device=dev, seed=11, reference="motion_lib")
        agent.vec_env.obs_carries_over = carries
        agent.init_tensors()
        agent.obs = agent.env_reset()
        agent._tensors_ready = True
        agent.play_steps()
        return agent
    a = played(True)
    assert a._boot_shortcut is True
    fast = a.experience_buffer.flat("next_values").clone()
    a._boot_shortcut = False                                              # same rollout, every row through the critic
    a._bootstrap_values()
    full = a.experience_buffer.flat("next_values").clone()
    tol = 1e-5 * max(1.0, full.abs().max().item())
    assert (fast - full).abs().max().item() <= tol
    b = played(False)                                                     # an env that does not promise observation continuity
    assert b._boot_shortcut is False
    assert (b.experience_buffer.flat("next_values") - full).abs().max().item() <= tol


@pytest.mark.parametrize("name,reference,fused", [("cfg1", "motion_lib", True), ("cfg1", "recorded", True), ("cfg5_small", "motion_lib", True),
                                                  ("cfg3_small", "motion_lib", None), ("speed_z_small", "motion_lib", None)])
def test_rollout_records_observations_without_copy_launches(dev, name, reference, fused):
    """[r6] play_steps records every observation twice (obses[n] before the policy step, next_obses[n] after the env step:
    a2c_common.play_steps).  The normaliser pass and the env's step kernel write those rows themselves (pulse_rms_normalize_copy,
    pulse_im_step_args.obs_copy); PULSE_OBS_SINK=0 is the two-copies-per-step form.  Every tensor of the experience buffer must be
    bit-identical either way, and the fused form must really be the one that ran."""
    def played(enabled):
        torch.manual_seed(7)
        agent, _ = configs.make_agent(name, device=dev, seed=21, reference=reference)
        agent._obs_sink_enabled = enabled
        agent.init_tensors()
        agent.obs = agent.env_reset()
        agent._tensors_ready = True
        calls = []
        eb = agent.experience_buffer
        orig = eb.update_data
        eb.update_data = lambda nm, i, v: (calls.append(nm), orig(nm, i, v))[1]
        batch = agent.play_steps()
        return agent, calls, batch
    a, calls_a, batch_a = played(True)
    b, calls_b, batch_b = played(False)
    assert calls_b.count("obses") == b.horizon_length and calls_b.count("next_obses") == b.horizon_length
    if fused:
        assert "obses" not in calls_a and "next_obses" not in calls_a, "the fused records fell back to copies"
        assert a.vec_env.obs_sink_written()
    for k in a.experience_buffer.phys:
        assert torch.equal(a.experience_buffer.phys[k], b.experience_buffer.phys[k]), k
    for k in ("returns", "advs_raw"):
        assert torch.equal(batch_a[k], batch_b[k]), k
    assert a.experience_buffer.phys["dones"].sum() > 0                    # resets happened: obses[n + 1] != next_obses[n] on those rows
    nx, ob = a.experience_buffer.phys["next_obses"], a.experience_buffer.phys["obses"]
    assert not torch.equal(nx[:, :-1], ob[:, 1:])


def test_obs_sink_is_refused_when_the_observation_is_post_processed(dev):
    """add_obs_noise / fut_tracks_dropout edit the row AFTER the step kernel (training time): the env must refuse the sink and the agent copies."""
    env, _ = configs.make_env(32, 8, dev, seed=3, reference="motion_lib", env_overrides={"add_obs_noise": True})
    env.alias_obs = True
    rows = torch.zeros(32, env.task.obs_pitch, device=dev)
    assert env.set_obs_sink(rows) is False
    env.reset()
    env.step(torch.zeros(32, 69, device=dev))
    assert not env.obs_sink_written() and rows.abs().sum() == 0
    env2, _ = configs.make_env(32, 8, dev, seed=3, reference="motion_lib")
    assert env2.set_obs_sink(rows) is False                               # a caller that may keep the returned tensor gets the reference's fresh copy
    env2.alias_obs = True
    assert env2.set_obs_sink(rows) is True
    env2.reset()
    obs, *_ = env2.step(torch.zeros(32, 69, device=dev))
    assert env2.obs_sink_written() and torch.equal(rows[:, :obs.shape[1]], obs)
    env2.step(torch.zeros(32, 69, device=dev))                            # the sink is one-shot
    assert not env2.obs_sink_written()
