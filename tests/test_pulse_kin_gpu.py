"""GPU: PULSE distillation step (AMPAgent._optimize_kin: action RMSE + KL(q || learned prior) + AR(1)) and the
amp_z PPO path, against the torch-CPU restatement; plus end-to-end epochs of both agent modes."""
import numpy as np
import pytest
import torch

from oracle import agent_oracle as AO
from pulse_amd import configs

pytestmark = pytest.mark.gpu


def rel_close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def test_optimize_kin_matches_reference_formulas(dev):
    torch.manual_seed(5)
    agent, _ = configs.make_agent("cfg3_small", device=str(dev), seed=5)
    agent.init_tensors()
    ref = AO.OracleNetZ()
    agent.model.load_state_dict(ref.state_dict_ref())
    t, mb = agent.horizon_length, agent.minibatch_size
    obs = torch.randn(mb, 934).clamp(-5, 5)
    gt = (0.4 * torch.randn(mb, 69)).clamp(-1, 1)
    prog = (torch.randint(0, 40, (mb // t, 1)) + torch.arange(t)[None, :]).reshape(-1, 1)
    prog[3 * t + 5:3 * t + t] = torch.arange(t - 5)[:, None]                  # an episode seam inside sequence 3 (reset -> progress restarts)
    noise = torch.randn(mb, 32)
    info_ref = AO.oracle_optimize_kin(ref, obs, gt, prog, noise, t)
    ws = agent.model.workspace(mb, train=True)
    ws["x"].zero_()
    ws["x"][:, :934] = obs.to(dev)
    agent.z_noise_provider = lambda m: noise.to(dev)
    agent.set_train()
    info = agent._optimize_kin(ws, mb, {"gt_action": gt.to(dev), "progress_buf": prog.to(dev)})
    for k in ("kin_action_loss", "kin_KLD", "kin_ar1", "kin_loss"):
        np.testing.assert_allclose(info[k].item(), info_ref[k].item(), rtol=2e-5, err_msg=k)
    grads = agent.model.net.gradients()
    total_sq = 0.0
    for name, p in ref.named_parameters():
        if name == "sigma":
            continue
        if p.grad is None:
            assert torch.count_nonzero(grads["a2c_network." + name]) == 0, f"{name} must get no gradient in kin mode"
            continue
        rel_close(grads["a2c_network." + name], p.grad, 3e-4, f"grad {name}")
        total_sq += p.grad.double().pow(2).sum().item()
    np.testing.assert_allclose(info["grad_norm"].item(), total_sq ** 0.5, rtol=1e-4)     # grad-norm within 1e-4


@pytest.mark.parametrize("name", ["cfg3_small", "cfg3_ppo_small"])
def test_amp_z_agent_epochs_run_and_learn(dev, name):
    agent, _ = configs.make_agent(name, device=str(dev), seed=3)
    before = agent.model.flat.clone()
    first = None
    for e in range(3):
        info = agent.train_epoch()
        key = "kin_loss" if "kin_loss" in info else "critic_loss"
        val = torch.stack(info[key]).mean().item()
        assert np.isfinite(val)
        first = val if first is None else first
    assert val < first, f"{key} should decrease over 3 epochs ({first} -> {val})"
    assert not torch.equal(before, agent.model.flat)
    eb = agent.experience_buffer
    assert torch.isfinite(eb.tensor_dict["mus"]).all() and torch.isfinite(eb.tensor_dict["values"]).all()
    if name == "cfg3_small":
        # kin mode: the env was stepped with mus; the critic never receives a gradient
        g = agent.model.net.gradients()
        assert torch.count_nonzero(g["a2c_network.critic_mlp.0.weight"]) == 0
        assert eb.tensor_dict["kin_dict"].shape[-1] == 70                      # gt_action (69) + progress_buf (1)


def test_humanoid_z_decoder_in_env(dev):
    """Downstream mode: latent action -> frozen prior + decoder -> 69-d PD action inside env.step."""
    from pulse_amd.env.humanoid_z import HumanoidImZ
    from pulse_amd.env.sim import RecordedMotion, RecordedRollout, RecordedSim
    torch.manual_seed(9)
    n = 130
    rollout = RecordedRollout(n, 3, seed=21).to(dev)
    sim = RecordedSim(rollout)
    task = HumanoidImZ({"env": dict(configs.ENV_IM, embedding_size=32)}, sim, RecordedMotion(rollout, sim), device=dev)
    assert task.num_actions == 32
    ref = AO.OracleNetZ()
    rm, rv = torch.randn(934).double() * 0.3, (torch.rand(934) + 0.5).double()
    task.initialize_z_models({"model": ref.state_dict_ref(), "running_mean_std": {"running_mean": rm, "running_var": rv}}, configs.NETWORK_Z)
    task.reset()
    az = 0.5 * torch.randn(n, 32)
    act = task.compute_z_actions(az.to(dev)).clone()
    want = AO.oracle_compute_z_actions(ref, task.obs_buf.cpu(), rm, rv, az)
    rel_close(act, want, 5e-5, "z -> action")
    task.step(az.to(dev))                                                       # full step with the latent action
    assert torch.isfinite(task.obs_buf).all() and task.rew_buf.shape == (n,)


@pytest.mark.parametrize("name", ["cfg3_small", "cfg3_ppo_small", "terrain_z_small"])
def test_region_reduce_equals_zero_fill_plus_uniform_reduce(dev, name, monkeypatch):
    """[r6] ParamBook.reduce_grads goes region by region (each parameter's own slab count, zeros over the sub-networks a pass does not
    visit, the norm clip's sums of squares from the same launch) instead of zero-filling the unvisited slabs and summing split_k slabs of
    everything.  Same summation order per element: the GRADIENT is bit-identical; the clip's norm is summed in another order, so the
    parameters after two epochs agree to round-off."""
    from pulse_amd.learning import graph as G

    def run(region):
        monkeypatch.setattr(G, "REGION_REDUCE", region)
        torch.manual_seed(4)
        agent, _ = configs.make_agent(name, device=str(dev), seed=9)
        agent.train_epoch()
        grad = agent.model.grad.clone()
        agent.train_epoch()
        return agent, grad
    a, ga = run(True)
    b, gb = run(False)
    book = a.model.book
    assert book.__dict__.get("_reduce_cache") and all(book._reduce_cache.values()), "the region reduce was not the one that ran"
    assert not b.model.book.__dict__.get("_reduce_cache")
    regs = [r for rg in book._reduce_cache for r in book.reduce_regions(rg)]
    assert any(ns == 0 for _, _, ns in regs) or name == "terrain_z_small"     # (the terrain policy visits every layer)
    d = (ga - gb).abs().max().item()
    assert d <= 1e-6 * max(1.0, gb.abs().max().item()), d          # (first epoch's last minibatch: parameters already differ by the clip's round-off)
    rel = (a.model.flat - b.model.flat).abs().max().item() / b.model.flat.abs().max().item()
    assert rel <= 1e-5, rel
