"""GPU: the trajectory / terrain task (pulse_traj_step, pulse_traj_generate; HumanoidTraj / HumanoidPedestrianTerrain(Z)) and the amp_sept
network against the goldens written by the reference's own code (tests/golden/terrain.npz, oracle/gen_golden.py: gen_terrain) and the
CPU oracle (oracle/task_oracle.py, oracle/agent_oracle.py: OracleNetSept).  Floats 1e-5, resets bit-exact (round-2 verdict, missing #1)."""
import os

import numpy as np
import pytest
import torch

from oracle import agent_oracle as AO
from oracle import task_oracle as TO
from pulse_amd import configs, ops, synthetic as syn
from pulse_amd._lib import TASK_OBS, TASK_RESET, TASK_REWARD

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "terrain.npz"))


def g(k, dev):
    return torch.from_numpy(Z[k]).to(dev)


def explain_height_differences(got, want, rb, upright, tol=1e-5, hs=None, hp=None, cp=None):
    """The height-map gather is index work: a device sample may only differ from the reference's where the sample's world point sits on a
    cell EDGE (the device's sinf / cosf / atan2f of the heading differ from the CPU libm by an ulp, so the rotated point moves by a few
    ulps and truncation picks the neighbouring cell).  For every differing sample this proves exactly that: (a) the point's cell
    coordinate is within 8 ulps of an integer in x or y, and (b) the device's value is the reference formula evaluated with the cell index
    shifted by one across that edge.  Returns the number of such samples; raises on any difference it cannot explain."""
    pad3 = lambda t: torch.cat([t, torch.zeros(t.shape[0], 1)], 1) if t.shape[1] == 2 else t
    hs = torch.from_numpy(Z["heightsamples"]) if hs is None else hs
    hp = pad3(torch.from_numpy(Z["height_points"]) if hp is None else hp)
    cp = pad3(torch.from_numpy(Z["center_points"]) if cp is None else cp)
    rbc = rb.cpu()
    head = syn.SMPL_BODY_NAMES.index("Head")
    sensor, root = rbc[:, head, 0:7], rbc[:, 0]
    pts = TO.terrain_sample_points(sensor, hp, upright)                      # (n, 1024, 3) fp32, the reference's arithmetic
    u = pts[..., :2] / 0.1                                                   # cell coordinates as the reference forms them
    center = TO.terrain_center_heights(hs, root, cp, 0.1, 0.005, upright).mean(dim=-1, keepdim=True)
    bad = ((got.cpu() - want.cpu()).abs() > tol).nonzero()
    explained = 0
    for e, k in bad.tolist():
        ux, uy = u[e, k, 0].item(), u[e, k, 1].item()
        near = [abs(c - round(c)) <= 8 * np.spacing(np.float32(abs(c))) for c in (ux, uy)]
        assert any(near), f"env {e} sample {k}: value differs but the point ({ux!r}, {uy!r}) is not on a cell edge"
        ix, iy = int(ux), int(uy)
        cands = []
        for dx in ((-1, 0, 1) if near[0] else (0,)):
            for dy in ((-1, 0, 1) if near[1] else (0,)):
                px = min(max(ix + dx, 0), hs.shape[0] - 2)
                py = min(max(iy + dy, 0), hs.shape[1] - 2)
                h = min(int(hs[px, py]), int(hs[px + 1, py + 1])) * 0.005
                cands.append(float(np.clip(np.float32(center[e, 0].item()) - np.float32(h), -3, 3) * 5.0))
        assert min(abs(got[e, k].item() - c) for c in cands) <= 2e-5, f"env {e} sample {k}: {got[e, k].item()} matches no neighbouring cell {cands}"
        explained += 1
    return explained


def test_traj_generate_matches_reference_trajectories(dev):
    rb, n, V = g("rb", dev), Z["verts"].shape[0], Z["verts"].shape[1]
    verts = torch.full((n, V, 3), float("nan"), device=dev)
    ops.traj_generate(rb, verts, g("u_dtheta", dev), g("u_sharp", dev), g("u_sharp_mask", dev), g("u_heading", dev), g("u_dspeed", dev),
                      g("u_speed0", dev), episode_dur=300 * (2.0 / 60.0))
    assert (verts - g("verts", dev)).abs().max().item() <= 2e-5            # 100-step prefix sums of cos / sin products
    # masked regeneration leaves the other envs alone
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    mask[::3] = True
    v2 = torch.zeros(n, V, 3, device=dev)
    ops.traj_generate(rb, v2, g("u_dtheta", dev), g("u_sharp", dev), g("u_sharp_mask", dev), g("u_heading", dev), g("u_dspeed", dev), g("u_speed0", dev),
                      episode_dur=300 * (2.0 / 60.0), env_mask=mask)
    assert torch.equal(v2[mask], verts[mask]) and (v2[~mask] == 0).all()


@pytest.mark.parametrize("upright", [True, False])
def test_terrain_task_observation_reward_reset_vs_golden(dev, upright):
    tag = "" if upright else "_noup"
    rb, verts, prog = g("rb", dev), g("verts", dev), g("progress", dev)
    n = rb.shape[0]
    common = dict(dt=float(Z["dt"]), episode_dur=300 * (2.0 / 60.0), upright=upright)
    terrain = dict(heightsamples=g("heightsamples", dev), height_points=g("height_points", dev), center_points=g("center_points", dev),
                   sensor_body=syn.SMPL_BODY_NAMES.index("Head"), use_center_height=True)
    obs = torch.full((n, 360 + 1044 + 4), float("nan"), device=dev)
    ops.traj_step(rb, verts, prog, what=TASK_OBS, obs=obs, obs_offset=360, **common, **terrain)
    got, want = obs[:, 360:360 + 1044], g(f"task_obs{tag}", dev)
    assert (got[:, :20] - want[:, :20]).abs().max().item() <= 1e-5          # trajectory samples in the heading frame
    # index work: every height sample equals the reference's to 1e-5 EXCEPT samples whose world point lies on a cell edge, and each of
    # those is proven to be the neighbouring cell's value (explain_height_differences raises on anything else); the count is reported
    flips = explain_height_differences(got[:, 20:], want[:, 20:], rb, upright)
    total = got[:, 20:].numel()
    print(f"[terrain] upright={upright}: {flips} of {total} height samples sit on a cell edge and take the neighbouring cell")
    assert flips <= 2e-3 * total
    assert torch.isnan(obs[:, :360]).all() and torch.isnan(obs[:, 360 + 1044:]).all()
    if upright:
        o2 = ops.traj_step(rb, verts, prog, what=TASK_OBS, **common)["obs"]                      # HumanoidTraj: no terrain observation
        assert (o2[:, :20] - g("traj_loc_obs", dev)).abs().max().item() <= 1e-5
        plane = ops.traj_step(rb, verts, prog, what=TASK_OBS, **common, height_points=g("height_points", dev), center_points=g("center_points", dev),
                              use_center_height=True)["obs"]
        assert (plane[:, 20:1044] == 0).all()                                               # terrainType 'plane': zero heights
    # reward: location term (+ power term in reward_raw)
    df, dv = torch.randn(n, 69, device=dev) * 50, torch.randn(n, 69, device=dev)
    out = ops.traj_step(rb, verts, prog, what=TASK_REWARD, dof_force=df, dof_vel=dv, power_coef=0.0005, power_reward=False, **common)
    assert (out["rew"] - g("loc_rew", dev)).abs().max().item() <= 1e-5
    pr = -0.0005 * (df * dv).abs().sum(-1)
    assert (out["rew_raw"][:, 0] - g("loc_rew", dev)).abs().max().item() <= 1e-5 and (out["rew_raw"][:, 1] - pr).abs().max().item() <= 1e-5 * pr.abs().max().item()
    out = ops.traj_step(rb, verts, prog, what=TASK_REWARD, dof_force=df, dof_vel=dv, power_coef=0.0005, power_reward=True, **common)
    assert (out["rew"] - (g("loc_rew", dev) + pr)).abs().max().item() <= 2e-5


def test_terrain_and_traj_reset_bit_exact(dev):
    rb = g("rb", dev).clone()
    rb[..., 0:3] = g("body_pos", dev)
    n = rb.shape[0]
    # the goldens use a displaced target: rebuild trajectories whose position at the env's time IS far_tar_pos (two-vertex tables)
    far = g("far_tar_pos", dev)
    verts = far[:, None, :].repeat(1, 2, 1).contiguous()
    prog = g("progress", dev)
    kw = dict(dt=float(Z["dt"]), episode_dur=10.0, contact_forces=g("contact", dev), contact_body_ids=g("contact_ids", dev),
              termination_heights=g("term_h", dev), max_episode_length=300.0, fail_dist=4.0)
    out = ops.traj_step(rb, verts, prog, what=TASK_RESET, terrain_reset=True, **kw)
    assert torch.equal(out["reset"], g("terrain_reset", dev)) and torch.equal(out["terminate"], g("terrain_terminated", dev))
    out = ops.traj_step(rb, verts, prog, what=TASK_RESET, terrain_reset=True, enable_early_termination=False, **kw)
    assert torch.equal(out["reset"], g("terrain_reset_noearly", dev)) and (out["terminate"] == 0).all()
    out = ops.traj_step(rb, verts, prog, what=TASK_RESET, terrain_reset=False, **kw)
    assert torch.equal(out["reset"], g("traj_reset", dev)) and torch.equal(out["terminate"], g("traj_terminated", dev))


def test_amp_sept_forward_and_gradients(dev):
    from pulse_amd.learning.network_sept import AMPSeptModel
    torch.manual_seed(6)
    m = 300
    ref = AO.OracleNetSept()
    model = AMPSeptModel(configs.NETWORK_SEPT, actions_num=32, self_obs_size=358, task_obs_size=1044, task_obs_size_detail={"traj": 20, "heightmap": 1024},
                         device=dev)
    sd = ref.state_dict_ref()
    model.load_state_dict(sd)
    back = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k
    obs = torch.randn(m, 1402).clamp(-5, 5)
    ws = model.workspace(m, train=True)
    ws["x"].zero_()
    ws["x"][:, :1402] = obs.to(dev)
    model.forward(ws, m)
    mu_r, _ = ref.eval_actor(obs)
    val_r = ref.eval_critic(obs)
    rel = lambda a, b: (a.detach().cpu().double() - b.detach().double()).abs().max().item() / (b.detach().double().abs().max().item() + 1e-12)
    assert rel(ws["mu"], mu_r) <= 5e-5 and rel(ws["val"], val_r) <= 5e-5
    wm, wv = torch.randn(m, 32), torch.randn(m, 1)
    ((mu_r * wm).sum() + (val_r * wv).sum()).backward()
    ws["dmu"].copy_(wm.to(dev))
    ws["dval"].copy_(wv.to(dev))
    model.backward(ws, m)
    grads = model.net.gradients()
    for name, p in ref.named_parameters():
        if p.grad is None:
            continue
        assert rel(grads["a2c_network." + name].reshape(p.grad.shape), p.grad) <= 3e-4, name
    # the shared task MLP got BOTH gradients: with only the critic seeded it still moves
    model.book.grad.zero_()
    ws["dmu"].zero_()
    model.backward(ws, m)
    assert model.net.gradients()["a2c_network._task_mlp.0.weight"].abs().sum().item() > 0


def test_terrain_z_env_steps_and_trains_end_to_end(dev):
    agent, _ = configs.make_agent("terrain_z_small", device=str(dev), seed=5)
    task = agent.vec_env.env.task
    assert task.get_task_obs_size_detail() == {"traj": 20, "heightmap": 1024} and task.num_obs == 358 + 1044 and task.num_actions == 32
    agent.init_tensors()
    agent.obs = agent.env_reset()
    obs = task.obs_buf
    assert torch.isfinite(obs).all() and obs[:, 378:].abs().max().item() <= 15.0 and obs[:, 378:].std().item() > 0.1     # clip(+-3) x 5, a real relief
    # the env's task observation equals the CPU oracle's on the same state
    rbc = task.sim.rigid_body_state.cpu()
    want = TO.terrain_task_obs(rbc[:, 0], rbc[:, 13, 0:7], TO.fetch_traj_samples(task._traj_verts.cpu(), task.progress_buf.cpu(), task.dt,
                               task._episode_dur / 100, 10, 0.5), syn.synthetic_height_field(),
                               torch.cat([syn.square_height_points(), torch.zeros(1024, 1)], 1), torch.cat([syn.center_height_points(), torch.zeros(9, 1)], 1))
    d = (obs[:, 358:].cpu() - want).abs()
    assert (d[:, :20] <= 1e-5).all()
    flips = explain_height_differences(obs[:, 378:].cpu(), want[:, 20:], rbc, True, hs=syn.synthetic_height_field(), hp=syn.square_height_points(),
                                       cp=syn.center_height_points())
    print(f"[terrain] env state: {flips} of {want[:, 20:].numel()} height samples sit on a cell edge and take the neighbouring cell")
    assert flips <= 2e-3 * want[:, 20:].numel()
    first = None
    for _ in range(3):
        info = agent.train_epoch()
        loss = torch.stack(info["actor_loss"]).mean().item()
        assert loss == loss
        first = first if first is not None else agent.model.flat.clone()
    assert not torch.equal(first, agent.model.flat) and torch.isfinite(agent.model.flat).all()
    assert (task.reset_buf.sum() + task.progress_buf.sum()).item() > 0
