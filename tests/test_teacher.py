"""The frozen PHC teacher (PNN primitives + composer) that produces the PULSE distillation target inside env.step.

CPU (build container): oracle_pnn_teacher_action == the reference's load_pnn / load_mcp_mlp / PNN.forward + the gt_action expression of
HumanoidImDistill.step, bit for bit, on a synthetic checkpoint.  GPU: pulse_amd.learning.teacher.PnnTeacher == the oracle, and the
teacher's action is what HumanoidIm hands to the agent as kin_dict['gt_action']."""
import numpy as np
import pytest
import torch

from oracle import agent_oracle as AO
from oracle import refload


def synthetic_teacher_checkpoints(num_prim=3, in_dim=934, units=(96, 64), comp_units=(80, 48), seed=0, has_lateral=False):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g)
    pm, cm = {}, {}
    for k in range(num_prim):
        i = in_dim
        for li, u in enumerate(list(units) + [69]):
            pm[f"a2c_network.pnn.actors.{k}.{2 * li}.weight"] = rnd(u, i) / i ** 0.5
            pm[f"a2c_network.pnn.actors.{k}.{2 * li}.bias"] = 0.1 * rnd(u)
            i = u
    if has_lateral:                                                          # pnn.py:24-37: u[i][j] = [Linear(u0 -> u1), Linear(u_last -> out)], no biases
        for i in range(num_prim - 1):
            for j in range(i + 1):
                pm[f"a2c_network.pnn.u.{i}.{j}.0.weight"] = rnd(units[1], units[0]) / units[0] ** 0.5
                pm[f"a2c_network.pnn.u.{i}.{j}.1.weight"] = rnd(69, units[-1]) / units[-1] ** 0.5
    pm["a2c_network.mu.weight"], pm["a2c_network.mu.bias"] = rnd(69, units[-1]), rnd(69)      # load_pnn reads the action size off mu.bias
    i = in_dim
    for li, u in enumerate(list(comp_units) + [num_prim]):
        cm[f"a2c_network.composer.{2 * li}.weight"] = rnd(u, i) / i ** 0.5
        cm[f"a2c_network.composer.{2 * li}.bias"] = 0.1 * rnd(u)
        i = u
    rms = {"running_mean": (0.3 * rnd(in_dim)).double(), "running_var": (torch.rand(in_dim, generator=g) + 0.5).double(), "count": torch.tensor(1e4).double()}
    return {"model": pm, "running_mean_std": rms}, {"model": cm}


@pytest.mark.skipif(not refload.available(), reason="reference checkout not mounted")
@pytest.mark.parametrize("activation", ["silu", "relu"])
@pytest.mark.parametrize("has_lateral", [False, True])
def test_oracle_teacher_matches_reference_loaders_and_pnn(activation, has_lateral):
    ref = refload.pnn_reference()
    pck, cck = synthetic_teacher_checkpoints(seed=3, has_lateral=has_lateral)
    pnn = ref["load_pnn"](pck, num_prim=3, has_lateral=has_lateral, activation=activation, device="cpu")
    composer = ref["load_mcp_mlp"](cck, activation=activation, device="cpu", mlp_name="composer")
    obs = torch.randn(40, 934, generator=torch.Generator().manual_seed(1)) * 2
    rm, rv = pck["running_mean_std"]["running_mean"], pck["running_mean_std"]["running_var"]
    with torch.no_grad():                                                  # humanoid_im_distill.py:165-198 with identical teacher / student obs
        full_obs = torch.clamp((obs - rm.float()) / torch.sqrt(rv.float() + 1e-05), min=-5.0, max=5.0)
        _, pnn_actions = pnn(full_obs)
        x_all = torch.stack(pnn_actions, dim=1)
        weights = composer(full_obs)
        want = torch.sum(weights[:, :, None] * x_all, dim=1)
    got = AO.oracle_pnn_teacher_action(pck["model"], cck["model"], 3, activation, obs, rm, rv, has_lateral=has_lateral)
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("activation,n,has_lateral", [("silu", 130, False), ("relu", 64, False), ("silu", 77, True), ("relu", 200, True)])
def test_teacher_matches_oracle(dev, activation, n, has_lateral):
    from pulse_amd.learning.teacher import PnnTeacher
    pck, cck = synthetic_teacher_checkpoints(seed=5, has_lateral=has_lateral)
    t = PnnTeacher(pck, cck, num_prim=3, num_envs=n, activation=activation, has_lateral=has_lateral, device=dev)
    obs = torch.randn(n, 960) * 2
    got = t.forward(obs.to(dev))
    want = AO.oracle_pnn_teacher_action(pck["model"], cck["model"], 3, activation, obs[:, :934], pck["running_mean_std"]["running_mean"],
                                        pck["running_mean_std"]["running_var"], has_lateral=has_lateral)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-5, rtol=2e-5)
    assert t.book.grad is None                                             # frozen: parameters only


@pytest.mark.gpu
def test_env_hands_the_teacher_action_to_the_agent(dev):
    from pulse_amd import configs
    from pulse_amd.learning.teacher import PnnTeacher
    ag, _ = configs.make_agent("cfg3_small", device=dev, seed=3, reference="motion_lib")
    task = ag.vec_env.env.task
    pck, cck = synthetic_teacher_checkpoints(seed=7)
    task.attach_teacher(PnnTeacher(pck, cck, num_prim=3, num_envs=task.num_envs, activation="silu", device=dev))
    ag.init_tensors()
    ag.obs = ag.env_reset()
    obs_before = task.obs_buf.clone()
    task.step(torch.zeros(task.num_envs, 69, device=dev))
    want = AO.oracle_pnn_teacher_action(pck["model"], cck["model"], 3, "silu", obs_before.cpu(), pck["running_mean_std"]["running_mean"],
                                        pck["running_mean_std"]["running_var"])
    np.testing.assert_allclose(task.extras["kin_dict"]["gt_action"].cpu().numpy(), want.numpy(), atol=2e-5, rtol=2e-5)
    info = ag.train_epoch()                                                # the distillation loss now regresses onto the teacher
    assert torch.isfinite(torch.stack([torch.as_tensor(v) for v in info["kin_action_loss"]])).all()
