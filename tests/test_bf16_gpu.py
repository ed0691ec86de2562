"""GPU: the bf16 MFMA path (BASELINE.json configs[4]: "AMP discriminator + PPO ... bf16").

Kernel level: pulse_gemm_f32 with compute_type BF16 (operands rounded to bf16 on the way into LDS, fp32 accumulation on
v_mfma_f32_32x32x16_bf16) against the same product formed from bf16-rounded operands in fp64 -- the three layouts, epilogues,
split-K, bias-gradient row sums, ragged sizes.
Agent level: AMPAgent with mixed_precision on cfg5_small against OracleAMPAgent under torch.autocast("cpu", bfloat16) (the
reference's autocast site: phc/learning/amp_agent.py:671).  The tolerance is DERIVED FROM THE ORACLE: the device result has to be
closer to the bf16 oracle than a fixed fraction of the distance between the bf16 oracle and the fp32 oracle."""
import numpy as np
import pytest
import torch

from oracle import amp_oracle as AMPO
from pulse_amd import configs, kernels as K
from pulse_amd._lib import ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG

pytestmark = pytest.mark.gpu

# floor of the first-step bound, relative to the quantity: 2 x the largest first-step |device - oracle16| / scale observed on MI355X over the six
# reported quantities (printed by the tests; round-4 verdict, weak #8: the old 4e-3 floor told little)
BF16_FLOOR = 1e-4          # (grad_norm 1.9e-5 here, 4.1e-5 at cfg5 full size: profiles/r05_bf16_parity_errors.txt; was 4e-3)


def bf(x):
    return x.bfloat16().double()


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (130, 70, 934), (517, 129, 70), (128, 128, 31), (4096, 512, 1024)])
def test_bf16_forward_gemm(dev, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    kp = (k + 3) // 4 * 4
    x = torch.zeros(m, kp)
    w = torch.zeros(n, kp)
    x[:, :k], w[:, :k] = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    out = torch.empty(m, n + 4 - n % 4 if n % 4 else n, device=dev)
    K.gemm(x.to(dev), w.to(dev), out, M=m, N=n, K=k, lda=kp, ldb=kp, ldc=out.stride(0), bias=b.to(dev), activation=ACT_RELU, compute_bf16=True)
    ref = torch.relu((bf(x[:, :k]) @ bf(w[:, :k]).T + b.double()).float().bfloat16().float())    # output rounded to bf16, then relu
    got = out[:, :n].cpu()
    assert torch.equal(got, got.bfloat16().float())                                   # outputs are bf16-representable
    err = (got.double() - ref.double()).abs()
    tol = 2.0 ** -7 * ref.abs().double() + 1e-6                                        # one bf16 ulp (a sum that lands on a rounding boundary may flip)
    assert (err <= tol).all(), float((err - tol).max())
    assert (err > 1e-7).double().mean() < 0.02                                         # ... and almost every element is exact


def test_bf16_dx_and_dw_gemms(dev):
    g = torch.Generator().manual_seed(5)
    m, n, k = 1000, 192, 136                       # dX: (m, k) = dY (m, n) W (n, k) ; dW: (n, k) = dY^T X
    dy, w, x = torch.randn(m, n, generator=g), torch.randn(n, k, generator=g) / n ** 0.5, torch.randn(m, k, generator=g)
    h = torch.randn(m, k, generator=g)
    dx = torch.empty(m, k, device=dev)
    K.gemm(dy.to(dev), w.to(dev), dx, M=m, N=k, K=n, lda=n, ldb=k, ldc=k, b_layout=GEMM_OUT_CONTIG, epilogue=EPI_RELU_GRAD, aux=h.to(dev), ldaux=k,
           compute_bf16=True)
    ref = ((bf(dy) @ bf(w)).float().bfloat16().float()) * (h > 0)
    err = (dx.cpu().double() - ref.double()).abs()
    assert (err <= 2.0 ** -7 * ref.abs().double() + 1e-6).all()
    S = 4
    slabs = torch.zeros(S, n * k + 256, device=dev)
    rows = torch.zeros(S, n * k + 256, device=dev)
    K.gemm(dy.to(dev), x.to(dev), slabs, M=n, N=k, K=m, lda=n, ldb=k, ldc=k, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, split_k=S,
           split_stride=slabs.stride(0), rowsum=rows, stride_rowsum=0, compute_bf16=True)
    dw = slabs.sum(0)[:n * k].view(n, k).cpu().double()
    ref = bf(dy).T @ bf(x)                                                              # split-K slabs are unrounded partial sums
    np.testing.assert_allclose(dw.numpy(), ref.numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(rows.sum(0)[:n].cpu().double().numpy(), bf(dy).sum(0).numpy(), rtol=2e-4, atol=2e-3)   # bias gradient of the bf16 operand


def _epoch(dev, seed, mixed):
    torch.manual_seed(seed)
    ag, _ = configs.make_agent("cfg5_small", device=str(dev), seed=seed, permutation_device="cpu", mixed_precision=mixed)
    ag.init_tensors()
    ag.obs = ag.env_reset()
    ag._tensors_ready = True
    T, N, A = ag.horizon_length, ag.num_actors, ag.actions_num
    noise = torch.randn(1, T, N, A, generator=torch.Generator().manual_seed(seed))
    nd = noise.to(dev)
    ag.noise_provider = lambda e, s: nd[e, s]
    init = (ag.model.state_dict(), ag.disc.state_dict())
    idx_lists = []
    inner = ag.train_actor_critic

    def spy(d):
        idx_lists.append(d["idx"].detach().cpu().clone())
        return inner(d)
    ag.train_actor_critic = spy
    ag.epoch_num = 1
    info = ag.train_epoch()
    return ag, info, noise, init, idx_lists


def _oracle(ag, noise, init, idx_lists, mixed):
    cfg = dict(ag.config)
    cfg["mixed_precision"] = mixed
    orc = AMPO.OracleAMPAgent(cfg, ag.obs_shape[0], ag._amp_dim, init[0], init[1], cfg["network"]["mlp"]["units"], (ag.disc.u1, ag.disc.u2))
    td = ag.experience_buffer.tensor_dict
    w = ag._amp_dim
    rec = {k: td[k].cpu().clone() for k in ("obses", "next_obses", "rewards", "dones", "terminates")}
    rec["amp_obs"] = td["amp_obs"].cpu().clone()[..., :w]
    orc.play_recorded(rec, noise[0])
    orc.prepare_dataset()
    ds = ag.dataset.values_dict
    demo = ag._amp_obs_demo_buffer.data[ds["_amp_demo_idx"]][:, :w].cpu()
    return orc.update(idx_lists, demo, ag.experience_buffer.flat("amp_obs")[:, :w].cpu())


def test_amp_agent_bf16_epoch_vs_autocast_oracle(dev):
    ag, info, noise, init, idx = _epoch(dev, 17, mixed=True)
    assert ag.mixed_precision and ag.model.mixed_precision and ag.disc.mixed_precision
    o16 = _oracle(ag, noise, init, idx, mixed=True)
    o32 = _oracle(ag, noise, init, idx, mixed=False)
    st = lambda key: torch.stack([torch.as_tensor(t).float().reshape(()) for t in info[key]]).cpu().double().numpy()
    for key in ("actor_loss", "critic_loss", "b_loss", "disc_loss", "disc_grad_penalty", "grad_norm"):
        dev_v = st(key)
        a16 = np.array([float(x[key]) for x in o16])
        a32 = np.array([float(x[key]) for x in o32])
        gap = np.abs(a16 - a32)                                   # what bf16 costs on the CPU
        err = np.abs(dev_v - a16)                                 # device bf16 vs CPU bf16 autocast
        scale = np.abs(a16) + 1e-6
        # first step: same weights on both sides -- the device must sit within the bf16 rounding noise of the autocast oracle
        print(f"[bf16 parity, 64-env agent] {key}: first step |device - oracle16| / scale = {err[0] / scale[0]:.2e}, |oracle16 - oracle32| / scale = "
              f"{gap[0] / scale[0]:.2e}; epoch max {float((err / scale).max()):.2e} (gap {float((gap / scale).max()):.2e})")
        assert err[0] <= max(2.0 * gap[0], BF16_FLOOR * scale[0]), (key, err[0], gap[0], scale[0])
        # whole epoch (weights drift through 24 Adam steps): stay within a few percent of the autocast oracle
        assert (err <= np.maximum(4.0 * gap, 3e-2 * scale)).all(), (key, float((err / scale).max()))
    # bf16 really was used: the device differs from the fp32 oracle where bf16 matters
    assert np.abs(st("disc_loss") - np.array([float(x["disc_loss"]) for x in o32])).max() > 0


def test_bf16_rollout_inference_stays_fp32(dev):
    """get_action_values is not under autocast in the reference (common_agent.py:262-288): rollout mus / values of a mixed-precision
    agent equal those of the fp32 agent bit for bit on the first epoch."""
    a16, _, _, _, _ = _epoch(dev, 23, mixed=True)
    a32, _, _, _, _ = _epoch(dev, 23, mixed=False)
    t16, t32 = a16.experience_buffer.tensor_dict, a32.experience_buffer.tensor_dict
    for k in ("mus", "values", "next_values", "actions"):
        assert torch.equal(t16[k], t32[k]), k
