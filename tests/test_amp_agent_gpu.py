"""GPU: the adversarial-motion-prior path of AMPAgent (enable_disc): discriminator rewards, the discriminator part of
calc_gradients (three normaliser passes + _disc_loss gradients) against the CPU oracle, and whole epochs of cfg5_small."""
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


@pytest.fixture(scope="module")
def agent(dev):
    torch.manual_seed(11)
    ag, _ = configs.make_agent("cfg5_small", device=dev, seed=4321)
    ag.init_tensors()
    ag.obs = ag.env_reset()
    return ag


def _oracle_twin(ag):
    ref = AO.OracleDisc(ag._amp_dim, units=(ag.disc.u1, ag.disc.u2))
    ref.load_state_dict({k.replace("a2c_network.", ""): v.cpu() for k, v in ag.disc.state_dict().items()})
    rms = AO.OracleRunningMeanStd((ag._amp_dim,))
    rms.running_mean, rms.running_var, rms.count = (t.cpu().clone() for t in (ag._amp_input_mean_std.running_mean,
                                                                              ag._amp_input_mean_std.running_var,
                                                                              ag._amp_input_mean_std.count))
    return ref, rms


def test_demo_buffer_and_env_amp_obs(agent):
    ag = agent
    assert ag._amp_obs_demo_buffer.get_total_count() >= ag._amp_obs_demo_buffer.get_buffer_size()
    assert ag._amp_dim == 10 * ag.vec_env.env.task._num_amp_obs_per_step
    assert torch.isfinite(ag._amp_obs_demo_buffer.data).all()
    assert ag._amp_obs_demo_buffer.data.abs().sum() > 0


def test_disc_rewards_match_oracle(agent):
    ag = agent
    torch.manual_seed(0)
    # non-trivial statistics
    warm = torch.randn(512, ag._amp_dim, device=ag.ppo_device) * 2 + 0.3
    ag._amp_input_mean_std.train()
    ag._amp_input_mean_std.forward(warm)
    ref, rms = _oracle_twin(ag)
    x = torch.randn(ag.batch_size, ag._amp_pitch, device=ag.ppo_device)
    got = ag._calc_disc_rewards(x)
    want = AO.oracle_disc_rewards(ref, rms, x[:, :ag._amp_dim].cpu(), ag._disc_reward_scale)
    rel_close(got, want, 5e-5, "disc rewards")
    assert got.shape == (ag.batch_size, 1)


def test_extra_gradients_match_oracle(agent):
    ag = agent
    dev = ag.ppo_device
    torch.manual_seed(1)
    b, n = ag._amp_minibatch_size, ag.batch_size
    ref, rms = _oracle_twin(ag)
    store = torch.randn(n, ag._amp_pitch, device=dev) * 1.5
    ag._amp_obs_demo_buffer.data.copy_(torch.randn_like(ag._amp_obs_demo_buffer.data) * 0.7 + 0.2)
    replay = torch.randn(ag._amp_replay_buffer.get_buffer_size(), ag._amp_pitch, device=dev)
    ag._amp_replay_buffer.data.copy_(replay)
    demo_idx = torch.randint(0, ag._amp_obs_demo_buffer.get_buffer_size(), (n,), device=dev)
    rep_idx = torch.randint(0, replay.shape[0], (n,), device=dev)
    idx = torch.randperm(n, device=dev)[:ag.minibatch_size]
    d = {"_amp_store": store, "_amp_demo_idx": demo_idx, "_amp_replay_src": ag._amp_replay_buffer.data, "_amp_replay_idx": rep_idx}
    ag.set_train()
    info = ag._extra_gradients({"dataset": d, "idx": idx}, idx)
    # ---- oracle: AMPAgent.calc_gradients (:621-629) + _disc_loss, disc_coef * disc_loss backward
    sub = idx[:b].cpu()
    w = ag._amp_dim
    rms.train()
    xa = rms(store.cpu()[sub, :w])
    xr = rms(replay.cpu()[rep_idx.cpu()[sub], :w])
    xd = rms(ag._amp_obs_demo_buffer.data.cpu()[demo_idx.cpu()[sub], :w])
    want = AO.oracle_disc_loss(ref, xa, xr, xd, ag._disc_logit_reg, ag._disc_grad_penalty, ag._disc_weight_decay)
    (ag._disc_coef * want["disc_loss"]).backward()
    np.testing.assert_allclose(info["disc_loss"].item(), want["disc_loss"].item(), rtol=1e-4)
    np.testing.assert_allclose(info["disc_grad_penalty"].item(), want["disc_grad_penalty"].item(), rtol=1e-4)
    np.testing.assert_allclose(info["disc_agent_acc"].item(), want["disc_agent_acc"].item(), atol=2.0 / (2 * b))
    np.testing.assert_allclose(info["disc_demo_acc"].item(), want["disc_demo_acc"].item(), atol=2.0 / b)
    grads = ag.disc.gradients()
    for name, p in ref.named_parameters():
        rel_close(grads["a2c_network." + name].reshape(p.shape), p.grad, 5e-4, f"disc grad {name}")
    rel_close(ag._amp_input_mean_std.running_mean, rms.running_mean, 2e-6, "amp rms mean")
    rel_close(ag._amp_input_mean_std.running_var, rms.running_var, 2e-5, "amp rms var")
    assert ag._amp_input_mean_std._count_host == float(rms.count)


def test_amp_epochs_train_policy_and_discriminator(dev):
    torch.manual_seed(5)
    ag, _ = configs.make_agent("cfg5_small", device=dev, seed=99)
    disc0, pol0 = ag.disc.flat.clone(), ag.model.flat.clone()
    for e in range(3):
        ag.epoch_num = e + 1
        info = ag.train_epoch()
        for k in ("disc_loss", "disc_grad_penalty", "disc_agent_acc", "disc_demo_acc", "actor_loss", "critic_loss", "grad_norm"):
            v = torch.stack([torch.as_tensor(t, device=dev).float() for t in info[k]])
            assert torch.isfinite(v).all(), k
        t, n = ag.horizon_length, ag.num_actors
        assert info["disc_rewards"].shape == (n, t, 1)
        # combined reward = 0.5 task + 0.5 disc (amp_agent.py:1011-1016) feeds GAE
        eb = ag.experience_buffer
        want = 0.5 * eb.phys["rewards"] + 0.5 * info["disc_rewards"]
        assert torch.allclose(info["mb_rewards"], want)
        assert (info["disc_rewards"] >= 0).all() and (info["disc_rewards"] <= -np.log(1e-4) * ag._disc_reward_scale + 1e-4).all()
    assert ag._amp_replay_buffer.get_total_count() == 3 * ag.batch_size
    assert not torch.equal(disc0, ag.disc.flat) and not torch.equal(pol0, ag.model.flat)
    # the discriminator learns to separate the (different) synthetic demo and agent distributions
    acc = torch.stack([torch.as_tensor(a) for a in info["disc_demo_acc"]]).float().mean().item()
    assert 0.0 <= acc <= 1.0


def test_joint_grad_clip_covers_both_buffers(agent):
    ag = agent
    groups = ag._param_groups()
    assert len(groups) == 2 and groups[1][0] is ag.disc.flat
    ag.model.grad.fill_(0.01)
    ag.disc.grad.fill_(0.02)
    ag._apply_gradients()
    want = np.sqrt(ag.model.n_flat * 0.01 ** 2 + ag.disc.n_flat * 0.02 ** 2)
    np.testing.assert_allclose(ag._grad_norm.item(), want, rtol=1e-4)


def test_policy_network_beside_a_concurrent_chain_keeps_off_the_fp32_bit_masks(dev, monkeypatch):
    """[r6] tools/mask_contend_probe.py: the fp32 x3 relu-grad epilogue that reads the ReLU bit mask returns wrong values now and then while another
    stream's GEMM waves share the SIMD.  AMPAgent runs its discriminator chain on a side stream, so its policy network must not use the fp32 bit
    masks; with the chain inline (PULSE_DISC_STREAM=0) and in single-chain agents it does."""
    from pulse_amd import configs
    from pulse_amd.learning import network as N
    monkeypatch.setattr(N, "RELU_BITMASK_F32", True)
    monkeypatch.setattr(N, "RELU_BITMASK_F32_FORCE", False)
    ag, _ = configs.make_agent("cfg5_small", device=str(dev), seed=3)
    assert ag._side_stream() is not None and ag.model.concurrent_chain
    ag.train_epoch()
    ws = ag.model.workspace(ag.minibatch_size, train=True)
    assert "hmask" not in ws
    monkeypatch.setenv("PULSE_DISC_STREAM", "0")
    ag2, _ = configs.make_agent("cfg5_small", device=str(dev), seed=3)
    assert ag2._side_stream() is None and not ag2.model.concurrent_chain
    ag2.train_epoch()
    assert "hmask" in ag2.model.workspace(ag2.minibatch_size, train=True)
    ag3, _ = configs.make_agent("cfg1", device=str(dev), seed=3, reference="motion_lib")
    ag3.train_epoch()
    assert "hmask" in ag3.model.workspace(ag3.minibatch_size, train=True)
