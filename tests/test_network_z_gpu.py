"""GPU: the PULSE VAE network (amp_z) on the graph executor vs the torch-CPU restatement of
AMPZBuilder.Network -- forward of encoder / prior / decoder / critic and every parameter gradient."""
import numpy as np
import pytest
import torch

from oracle import agent_oracle as AO
from pulse_amd import configs
from pulse_amd.learning.network_z import AMPZNetwork

pytestmark = pytest.mark.gpu

NET_Z = {"separate": True,
         "space": {"continuous": {"sigma_init": {"name": "const_initializer", "val": -2.9}, "fixed_sigma": True, "learn_sigma": False}},
         "mlp": {"units": [3096, 2048, 1024], "activation": "silu"}, "task_mlp": {"units": [1536, 1024, 512], "activation": "silu"}}
DETAIL = {"embedding_size": 32, "z_type": "vae", "use_vae_prior": True, "use_vae_clamped_prior": True, "vae_var_clamp_max": 2}


def rel_close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def test_amp_z_forward_and_gradients(dev):
    torch.manual_seed(3)
    m = 300
    ref = AO.OracleNetZ()
    net = AMPZNetwork(NET_Z, actions_num=69, self_obs_size=358, task_obs_size=576, task_obs_size_detail=DETAIL, device=dev)
    sd = ref.state_dict_ref()
    net.load_state_dict(sd)
    back = net.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k                              # names / shapes / column maps round-trip exactly
    obs = torch.randn(m, 934).clamp(-5, 5)
    noise = torch.randn(m, 32)
    x = torch.zeros(m, 960, device=dev)
    x[:, :934] = obs.to(dev)
    G = net.graph(m, x)
    g = G["g"]
    # ---------------- forward
    G["fwd_enc"].run()
    G["fwd_prior"].run()
    vae_mu, vae_logvar = net.split_heads(g.act_bufs["zheads"])
    z = vae_mu + torch.exp(0.5 * vae_logvar) * noise.to(dev)
    g.act_bufs["ain"][:, :358] = x[:, :358]
    g.act_bufs["ain"][:, 360:392] = z
    g.act_bufs["cin"][:, :358] = x[:, :358]
    G["fwd_dec"].run()
    G["fwd_critic"].run()
    obs_r = obs.clone().requires_grad_(False)
    mu_r, _, extra = ref.eval_actor(obs_r, noise)
    pm_r, pv_r = ref.compute_prior(obs_r)
    val_r = ref.eval_critic(obs_r)
    rel_close(vae_mu, extra["vae_mu"], 2e-5, "vae_mu")
    rel_close(vae_logvar, extra["vae_log_var"], 2e-5, "vae_log_var")
    pm, pv = net.split_heads(g.act_bufs["pheads"])
    rel_close(pm, pm_r, 2e-5, "prior_mu")
    rel_close(pv, pv_r, 2e-5, "prior_logvar")
    rel_close(g.act_bufs["mu"][:, :69], mu_r, 5e-5, "mu")
    rel_close(g.act_bufs["value"][:, :1], val_r, 5e-5, "value")
    # ---------------- backward of a random linear functional of every output
    wm, we, wp, wv = torch.randn(m, 69), torch.randn(m, 64), torch.randn(m, 64), torch.randn(m, 1)
    # reference: heads feed the loss both directly (KL-like term `we`) and through z -> decoder
    loss = (mu_r * wm).sum() + (torch.cat([extra["vae_mu"], extra["vae_log_var"]], -1) * we).sum() + (torch.cat([pm_r, pv_r], -1) * wp).sum() \
        + (val_r * wv).sum()
    loss.backward()
    # device: seed head gradients, run decoder backward, route dz through the re-parameterisation with autograd on the tiny heads
    g.grad("mu")[:, :69] = wm.to(dev)
    g.grad("value")[:, :1] = wv.to(dev)
    G["bwd_dec"].run()
    G["bwd_critic"].run()
    dz = g.grad("ain")[:, 360:392]
    heads = g.act_bufs["zheads"].detach().clone().requires_grad_(True)
    hm, hv = net.split_heads(heads)
    zz = hm + torch.exp(0.5 * hv) * noise.to(dev)
    small = (torch.cat([hm, hv], -1) * we.to(dev)).sum()
    torch.autograd.backward([small, zz], [None, dz])
    g.grad("zheads").copy_(heads.grad)
    ph = g.act_bufs["pheads"].detach().clone().requires_grad_(True)
    pmm, pvv = net.split_heads(ph)
    (torch.cat([pmm, pvv], -1) * wp.to(dev)).sum().backward()
    g.grad("pheads").copy_(ph.grad)
    G["bwd_enc"].run()
    G["bwd_prior"].run()
    net.book.reduce_grads()
    grads = net.gradients()
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        rel_close(grads["a2c_network." + name], p.grad, 3e-4, f"grad {name}")
    # column-map gaps never receive weight or gradient
    w = net.book.phys("a2c_network.actor_mlp.0.weight")
    assert torch.count_nonzero(w[:, 358:360]) == 0
    assert torch.count_nonzero(net.book.phys("a2c_network.actor_mlp.0.weight", net.book.grad)[:, 358:360]) == 0


def test_amp_z_second_batch_size_shares_parameters(dev):
    net = AMPZNetwork(NET_Z, actions_num=69, self_obs_size=358, task_obs_size=576, task_obs_size_detail=DETAIL, device=dev)
    a, b = net.graph(64), net.graph(128)
    assert a["g"].book.flat.data_ptr() == b["g"].book.flat.data_ptr() == net.book.flat.data_ptr()
    a["x"].normal_()
    b["x"][:64] = a["x"]
    a["fwd_enc"].run()
    b["fwd_enc"].run()
    assert torch.equal(a["g"].act_bufs["zheads"], b["g"].act_bufs["zheads"][:64])
