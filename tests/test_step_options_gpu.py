"""GPU: four small options of HumanoidIm's step composition that no shipped config enables but that sit inside the line ranges SURVEY.md
section 8 cites (rows a9 / a11 / a13) -- cycle_motion_xp (phc/env/tasks/humanoid_im.py:1133-1134), fut_tracks_dropout (:804-810), add_obs_noise
(:691-692), res_action (:1096-1101).  The oracle's restatements are pinned to the reference's method bodies on the CPU
(tests/test_oracle_env_vs_reference_methods.py); here the device env is compared with them, the random draws replayed."""
import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from oracle.motion_oracle import OracleMotionEnv, OracleMotionLib
from pulse_amd import configs
from pulse_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _env(dev, n, overrides, seed=77):
    env, _ = configs.make_env(n, 24, dev, seed=seed, reference="motion_lib", env_overrides=overrides)
    return env, env.task


def test_cycle_motion_xp_lockstep(dev):
    n, seed = 52, 321
    env, task = _env(dev, n, {"cycle_motion": True, "cycle_motion_xp": True, "episode_length": 45}, seed)
    tabs = syn.synthetic_motion_library(syn.make_generator(seed + 5, 0), min(n, 1024))
    twin = OracleMotionEnv(OracleMotionLib(tabs), {k: v.cpu() for k, v in task.sim.bank.items()}, task._sampled_motion_ids.cpu(), task._global_offset.cpu(),
                           task._reset_bodies_id.cpu().long(), task._track_bodies_id.cpu().long(), task.dt, cycle_motion=True, max_episode_length=45,
                           cycle_motion_xp=True)
    obs = env.reset()
    o_ref = twin.reset(torch.arange(n), task._motion_start_times.cpu())
    np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5)
    for step in range(50):
        obs, rew, done, info = env.step(torch.zeros(n, 69, device=dev))
        o_ref, r_ref, d_ref, i_ref = twin.step(cycle_start_times=task._last_cycle_start.cpu(), xp_uniforms=task._last_xp_uniforms.cpu())
        np.testing.assert_allclose(rew.cpu().numpy(), r_ref.numpy(), atol=2e-5, rtol=1e-5)
        assert torch.equal(done.cpu(), d_ref)
        np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5, err_msg=f"obs step {step}")
        np.testing.assert_allclose(task._global_offset.cpu().numpy(), twin.offset.numpy(), atol=1e-5)
        ids = torch.nonzero(d_ref).flatten()
        env.reset(ids.to(dev))
        twin.reset(ids, task._motion_start_times.cpu())
    off = task._global_offset[:, 0:2]
    assert (off != 0).any()


def test_fut_tracks_dropout_and_obs_noise(dev):
    n, T = 40, 3
    base = {"fut_tracks": True, "numTrajSamples": T}
    env0, t0 = _env(dev, n, dict(base))
    env1, t1 = _env(dev, n, dict(base, fut_tracks_dropout=True, add_obs_noise=True))
    o0, o1 = env0.reset(), env1.reset()
    sw, per = t1._self_obs_size, t1._task_obs_size // T

    def check(clean, noisy, mask=None):
        u, z = t1._last_dropout_uniforms.cpu(), t1._last_obs_noise.cpu()
        want = torch.cat([clean[:, :sw].cpu(), E.fut_tracks_dropout(clean[:, sw:].cpu(), u, T)], dim=-1)
        want = E.add_obs_noise(want, z)
        if mask is not None:
            want = torch.where(mask[:, None].cpu(), want, noisy.cpu())
        np.testing.assert_allclose(noisy.cpu().numpy(), want.numpy(), atol=1e-6, rtol=0)
        return u

    u = check(o0, o1)
    assert (u < 0.1).any() and (u >= 0.1).any()
    dropped = (u < 0.1)
    blocks = o1[:, sw:].cpu().view(n, T, per) - t1._last_obs_noise.cpu()[:, sw:].view(n, T, per) * 0.1
    assert (blocks[dropped].abs() < 1e-6).all()
    for step in range(4):
        a = torch.zeros(n, 69, device=dev)
        o0, _, d0, _ = env0.step(a)
        o1, _, d1, _ = env1.step(a)
        assert torch.equal(d0, d1)                                   # rewards / resets do not see the observation options
        check(o0, o1)
        mask = d0 > 0
        if mask.any():
            prev = env1.task.obs_buf.clone()
            o0 = env0.reset_masked(mask)
            o1 = env1.reset_masked(mask)
            check(o0, o1, mask)
            assert torch.equal(o1[~mask], prev[~mask])               # envs that were not reset keep their (noisy) observation
    # flags.test: evaluation sees the clean observation
    t1.test = True
    o0, _, _, _ = env0.step(a)
    o1, _, _, _ = env1.step(a)
    assert torch.equal(o0, o1)


def test_res_action_pd_targets(dev):
    n = 33
    env, task = _env(dev, n, {"res_action": True})
    env.reset()
    seen = []
    real = task.sim.set_dof_position_target_tensor
    task.sim.set_dof_position_target_tensor = lambda t: (seen.append(t.clone()), real(t))[1]
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        act = (torch.randn(n, 69, generator=g) * 2.0).to(dev)
        ref_dof, dof = task._track["dof_pos"].clone(), task.sim.dof_pos.clone()
        env.step(act)
        want = E.res_action_pd_targets(ref_dof.cpu(), task._pd_action_scale.cpu(), act.clamp(-1, 1).cpu(), dof.cpu())
        np.testing.assert_allclose(seen[-1].cpu().numpy(), want.numpy(), atol=1e-6, rtol=0)
    with pytest.raises(NotImplementedError):
        configs.make_env(8, 4, dev, env_overrides={"res_action": True})      # recorded reference frames carry no reference dof positions


@pytest.mark.parametrize("overrides", [{}, {"obs_v": 7}, {"zero_out_far": True, "zero_out_far_train": False}])
def test_occl_training_lockstep(dev, overrides):
    """occl_training (humanoid_im.py:778-784, 827-831, 1178-1183): the fused step with a per-env occlusion word against the CPU twin (pinned to
    the reference's method bodies), a fresh RANDOM mask every step -- the generality the reference's data structure has -- and then the
    reference's own mask update, whose last two statements fix the mask to bodies 0 .. 8."""
    n, seed = 61, 321
    env, task = _env(dev, n, dict({"occl_training": True}, **overrides), seed)
    tabs = syn.synthetic_motion_library(syn.make_generator(seed + 5, 0), min(n, 1024))
    twin = OracleMotionEnv(OracleMotionLib(tabs), {k: v.cpu() for k, v in task.sim.bank.items()}, task._sampled_motion_ids.cpu(), task._global_offset.cpu(),
                           task._reset_bodies_id.cpu().long(), task._track_bodies_id.cpu().long(), task.dt, obs_v=task.obs_v,
                           zero_out_far=task.zero_out_far, zero_out_far_train=False)
    g = torch.Generator().manual_seed(5)
    real_update = task._update_occl_training

    def random_mask():
        m = torch.rand(n, 24, generator=g) < 0.3
        task.random_occlu_idx = m.to(dev)
        task._pack_occl_bits()
        twin.occl_idx = m

    task._update_occl_training = random_mask
    random_mask()
    obs = env.reset()
    o_ref = twin.reset(torch.arange(n), task._motion_start_times.cpu())
    np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5)
    dones = 0
    for step in range(30):
        obs, rew, done, info = env.step(torch.zeros(n, 69, device=dev))          # pre_physics_step installs the step's mask
        o_ref, r_ref, d_ref, i_ref = twin.step()
        np.testing.assert_allclose(rew.cpu().numpy(), r_ref.numpy(), atol=2e-5, rtol=1e-5)
        assert torch.equal(done.cpu(), d_ref) and torch.equal(info["terminate"].cpu(), i_ref["terminate"]), f"flags step {step}"
        np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5, err_msg=f"obs step {step}")
        ids = torch.nonzero(d_ref).flatten()
        dones += ids.numel()
        obs = env.reset(ids.to(dev))
        if ids.numel():
            o_ref = twin.reset(ids, task._motion_start_times.cpu())
            np.testing.assert_allclose(obs.cpu().numpy(), o_ref.numpy(), atol=5e-5, rtol=1e-5)
    assert dones > 0
    # an occluded body's difference blocks are exact zeros
    sw = task._self_obs_size
    dpos = obs[:, sw:sw + 72].cpu().view(n, 24, 3)
    assert (dpos[twin.occl_idx] == 0).all() and (dpos[~twin.occl_idx] != 0).any()
    # the reference's own update: whatever it draws, the mask it leaves is bodies 0 .. 8
    real_update()
    want = torch.zeros(n, 24, dtype=torch.bool)
    want[:, :9] = True
    assert torch.equal(task.random_occlu_idx.cpu(), want) and (task._occl_bits.cpu() == 0x1FF).all()
    assert (task.random_occlu_count >= 0).all() and (task.random_occlu_count[:, 0] == 0).all() and (task.random_occlu_count > 0).any()


def test_occl_training_needs_the_full_body(dev):
    with pytest.raises(NotImplementedError):
        configs.make_env(8, 4, dev, reference="motion_lib", env_overrides={"occl_training": True, "trackBodies": ["Head", "L_Hand", "R_Hand"], "obs_v": 7})


def test_add_amp_input_noise(dev):
    """humanoid_amp.py:281-283: demo windows + 0.01 N(0, 1)."""
    n = 16
    e0, t0 = configs.make_env(n, 8, dev, seed=77, env_kind="amp", reference="motion_lib")[0], None
    e1 = configs.make_env(n, 8, dev, seed=77, env_kind="amp", reference="motion_lib", env_overrides={"add_amp_input_noise": True})[0]
    e0.reset(); e1.reset()
    a, b = e0.fetch_amp_obs_demo(24), e1.fetch_amp_obs_demo(24)
    np.testing.assert_allclose(b.cpu().numpy(), (a + e1.task._last_amp_noise * 0.01).cpu().numpy(), atol=1e-6, rtol=0)
    assert (a != b).any()
