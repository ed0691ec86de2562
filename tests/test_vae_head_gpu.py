"""GPU: the fused PULSE VAE head kernels (pulse_amd/csrc/vae_head.hip) against the reference's formulas evaluated with torch autograd:
form_embedding (phc/learning/amp_network_z_builder.py:79-121), the losses of AMPAgent._optimize_kin (amp_agent.py:771-849; kl_multi
loss_functions.py:3-10) and every head-level gradient, including the clamp masks and the AR(1) seam mask."""
import pytest
import torch

from oracle import agent_oracle as AO
from pulse_amd import kernels as K

pytestmark = pytest.mark.gpu


def _inputs(dev, seqs=24, t=16, E=32, A=69, seed=0):
    g = torch.Generator().manual_seed(seed)
    mb = seqs * t
    zh = torch.randn(mb, 2 * E, generator=g)
    zh[:, E:] *= 3.0                                                        # raw log-variances on both sides of the clamp range [-5, 2]
    ph = torch.randn(mb, 2 * E, generator=g)
    ph[:, E:] *= 3.0
    pred, gt = torch.randn(mb, A, generator=g), torch.randn(mb, A, generator=g) * 0.4
    gt[3] = pred[3]                                                         # a zero-norm row: torch.norm's backward gives 0 there
    eps = torch.randn(mb, E, generator=g)
    dz = torch.randn(mb, E, generator=g) * 1e-3
    prog = (torch.randint(0, 50, (seqs, 1), generator=g) + torch.arange(t)[None, :])
    prog[2, 5:] = torch.arange(t - 5)                                        # an episode seam inside sequence 2
    prog[4, :] = torch.arange(t)                                             # a sequence that starts at frame 0 (first frames are masked)
    return [x.to(dev) for x in (zh, ph, pred, gt, eps, dz, prog.reshape(-1, 1))], mb, t, E, A


def test_embed_matches_form_embedding(dev):
    (zh, _, _, _, eps, _, _), mb, t, E, A = _inputs(dev)
    S, zc = 358, 360
    x = torch.randn(mb, 960, device=dev)
    ain = torch.full((mb, 392), float("nan"), device=dev)
    cin = torch.full((mb, 392), float("nan"), device=dev)
    K.vae_embed(zh, x, ain, rows=mb, embedding_size=E, self_obs_size=S, z_col=zc, eps=eps, cin=cin, clamp=True, clamp_max=2.0)
    z = zh[:, :E] + torch.exp(0.5 * torch.clamp(zh[:, E:], min=-5, max=2.0)) * eps
    assert torch.equal(ain[:, :S], x[:, :S]) and torch.equal(cin[:, :S], x[:, :S])
    assert (ain[:, zc:zc + E] - z).abs().max().item() <= 2e-6 * z.abs().max().item()
    assert torch.isnan(ain[:, S:zc]).all() and torch.isnan(cin[:, S:]).all()          # nothing else is touched
    K.vae_embed(zh, x, ain, rows=mb, embedding_size=E, self_obs_size=S, z_col=zc, eps=None)     # test mode: z = mu
    assert torch.equal(ain[:, zc:zc + E], zh[:, :E])


@pytest.mark.parametrize("use_ar1,use_regu", [(True, False), (True, True), (False, False)])
def test_kin_losses_and_head_gradients_match_autograd(dev, use_ar1, use_regu):
    (zh, ph, pred, gt, eps, dz, prog), mb, t, E, A = _inputs(dev, seed=3)
    kld_w, ar1_w = 0.01, 0.005
    # ---- the reference's formulas under autograd
    zh_r, ph_r, pred_r = zh.clone().requires_grad_(True), ph.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    qm, qv = zh_r[:, :E], torch.clamp(zh_r[:, E:], min=-5, max=2.0)
    pm, pv = ph_r[:, :E], torch.clamp(ph_r[:, E:], min=-5, max=2.0)
    z = qm + torch.exp(0.5 * qv) * eps
    act = torch.norm(pred_r - gt, dim=-1).mean()
    kld = AO.kl_multi(qm, qv, pm, pv).mean()
    ar1 = torch.zeros((), device=dev)
    if use_ar1:
        zs = qm.view(mb // t, t, -1)
        err = (zs[:, 1:] - zs[:, :-1] * 0.99).reshape(-1, E)
        idx = prog.view(mb // t, t, -1)
        bad = ((idx[:, 1:] - idx[:, :-1]) != 1).view(-1) | ((idx <= 2)[:, 1:] | (idx <= 2)[:, :-1]).view(-1)
        err = err * (~bad).float()[:, None]
        ar1 = torch.norm(err, dim=-1).mean()
    regu = ((pm ** 2).mean() + (qm ** 2).mean()) * 0.001 + ((pv ** 2).mean() + (qv ** 2).mean()) * 0.001 if use_regu else torch.zeros((), device=dev)
    loss = act + kld * kld_w + ar1 * ar1_w + regu * 0.005
    torch.autograd.backward([loss, z], [None, dz])
    # ---- the kernels
    dmu = torch.full((mb, A), float("nan"), device=dev)
    partials = torch.zeros(64, 8, device=dev)
    p1 = prog.reshape(-1).contiguous()
    K.vae_kin_loss(pred, gt, zh, ph, p1 if use_ar1 else None, dmu, partials, rows=mb, num_actions=A, embedding_size=E, horizon=t, use_ar1=use_ar1,
                   use_regu=use_regu)
    s = partials.sum(0)
    n_err = (mb // t) * (t - 1)
    close = lambda a, b, tol=2e-5: abs(float(a) - float(b)) <= tol * max(abs(float(b)), 1e-6)
    assert close(s[0] / mb, act) and close(s[1] / mb, kld)
    if use_ar1:
        assert close(s[2] / n_err, ar1)
    if use_regu:
        assert close((s[3] + s[4]) / (mb * E) * 0.001 + (s[5] + s[6]) / (mb * E) * 0.001, regu)
    assert (dmu - pred_r.grad).abs().max().item() <= 1e-6 * pred_r.grad.abs().max().item() + 1e-12 and (dmu[3] == 0).all()
    dzh = torch.full((mb, 2 * E), float("nan"), device=dev)
    dph = torch.full((mb, 2 * E), float("nan"), device=dev)
    K.vae_head_backward(zh, dzh, rows=mb, embedding_size=E, horizon=t, pheads=ph, dpheads=dph, eps=eps, dz=dz, progress=p1 if use_ar1 else None,
                        c_kl=kld_w / mb, c_ar1=(ar1_w / n_err) if use_ar1 else 0.0, c_regu=(0.005 * 0.001 / (mb * E)) if use_regu else 0.0)
    for got, want, name in ((dzh, zh_r.grad, "encoder heads"), (dph, ph_r.grad, "prior heads")):
        err = (got - want).abs().max().item()
        assert err <= 3e-6 * want.abs().max().item() + 1e-10, (name, err, want.abs().max().item())
    # clamp mask: no gradient reaches a raw log-variance outside [-5, 2]
    outside = (zh[:, E:] < -5) | (zh[:, E:] > 2)
    assert outside.any() and (dzh[:, E:][outside] == 0).all()


def test_head_backward_without_prior_is_the_reparameterisation_path(dev):
    (zh, _, _, _, eps, dz, _), mb, t, E, A = _inputs(dev, seed=5)
    dzh = torch.empty(mb, 2 * E, device=dev)
    K.vae_head_backward(zh, dzh, rows=mb, embedding_size=E, eps=eps, dz=dz)
    qv = torch.clamp(zh[:, E:], min=-5, max=2.0)
    inside = (zh[:, E:] >= -5) & (zh[:, E:] <= 2)
    want_v = dz * 0.5 * torch.exp(0.5 * qv) * eps * inside
    assert torch.equal(dzh[:, :E], dz) and (dzh[:, E:] - want_v).abs().max().item() <= 2e-6 * want_v.abs().max().item()
