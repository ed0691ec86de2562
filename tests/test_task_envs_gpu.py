"""GPU: the downstream-task environments (HumanoidSpeed / Reach / Strike and the latent-action Z forms) step through the HIP
kernels and agree with the CPU task oracle on the same simulated state; obs variants plumbed through HumanoidIm; a
HumanoidSpeedZ + amp_z_reader policy (learning=pulse_z_task shape) trains end to end."""
import numpy as np
import pytest
import torch

from oracle import env_oracle as E
from oracle import task_oracle as TO
from pulse_amd import configs, synthetic as syn
from pulse_amd.env import humanoid_tasks as HT

pytestmark = pytest.mark.gpu


def _cmp(a, b, atol=1e-5):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), atol=atol, rtol=1e-5)


@pytest.mark.parametrize("name", ["HumanoidSpeed", "HumanoidReach", "HumanoidStrike"])
def test_task_env_steps_match_oracle(dev, name):
    n, frames = 130, 12
    sim = HT.SyntheticTaskSim(n, frames, dev, seed=5)
    task = HT.TASKS[name]({"env": {"power_reward": name == "HumanoidSpeed", "episode_length": 8}}, sim, device=dev)
    task.reset()
    assert task.obs_buf.shape == (n, 358 + (15 if name == "HumanoidStrike" else 3))
    saw_reset = 0
    for step in range(10):
        prev_root = sim.rigid_body_state[:, 0, 0:3].clone().cpu()
        prog_before = task.progress_buf.clone().cpu()
        task.step(torch.zeros(n, 69, device=dev))
        rb = sim.rigid_body_state.cpu()
        root = rb[:, 0]
        bp, br, bv, ba = E.split_rb(rb)
        self_obs = E.self_obs_smpl_max(bp, br, bv, ba)
        prog = prog_before + 1
        if name == "HumanoidSpeed":
            tobs = TO.speed_observations(root, task._tar_speed.cpu())
            rew = TO.speed_reward(root[:, 0:3], prev_root, root[:, 3:7], task._tar_speed.cpu(), task.dt)
            pw = -0.0005 * (sim.dof_force * sim.dof_vel).abs().sum(-1).cpu()
            pw[prog <= 3] = 0
            rew = rew + pw
        elif name == "HumanoidReach":
            tobs = TO.location_observations(root, task._tar_pos.cpu())
            rew = TO.reach_reward(rb[:, task._reach_body_id, 0:3], root[:, 3:7], task._tar_pos.cpu(), 1.0, task.dt)
        else:
            ts = task._target_states.cpu()
            tobs = TO.strike_observations(root, ts)
            rew = TO.strike_reward(ts[:, 0:3], ts[:, 3:7], root, prev_root, rb[:, 23, 7:10], task.dt, 1.4)
        _cmp(task.obs_buf, torch.cat([self_obs, tobs], dim=-1))
        _cmp(task.rew_buf, rew)
        args = (torch.zeros(n, dtype=torch.long), prog, sim.contact_forces.cpu(), task._contact_body_ids.cpu().long(), rb[..., 0:3])
        if name == "HumanoidStrike":
            r, t = TO.strike_reset(*args, sim.target_contact_forces.cpu(), task._strike_body_ids.cpu().long(), float(task.max_episode_length), True,
                                   task._termination_heights.cpu())
        else:
            r, t = TO.humanoid_reset(*args, float(task.max_episode_length), True, task._termination_heights.cpu())
        assert torch.equal(task.reset_buf.cpu(), r) and torch.equal(task._terminate_buf.cpu(), t)
        saw_reset += int(r.sum())
        task.reset_masked(task.reset_buf.bool())
    assert saw_reset > 0


def test_speed_task_target_schedule(dev):
    n = 64
    sim = HT.SyntheticTaskSim(n, 8, dev, seed=2)
    task = HT.HumanoidSpeed({"env": {"speedChangeStepsMin": 2, "speedChangeStepsMax": 4, "tarSpeedMin": 1.0, "tarSpeedMax": 3.0, "episode_length": 1000}}, sim, device=dev)
    task.reset()
    first = task._tar_speed.clone()
    assert ((first >= 1.0) & (first <= 3.0)).all() and ((task._speed_change_steps >= 2) & (task._speed_change_steps < 4)).all()
    for _ in range(4):
        task.step(torch.zeros(n, 69, device=dev))
    assert (task._tar_speed != first).all()                        # every env passed its change step (< 4) and drew a new target
    # _update_task runs in pre_physics_step, BEFORE progress_buf advances (humanoid_amp_task.py:57-59): a target drawn at progress p is
    # due again at p + [min, max), so after the step the change step is at least the new progress
    assert (task._speed_change_steps >= task.progress_buf).all()


@pytest.mark.parametrize("over", [{"obs_v": 1}, {"obs_v": 3, "trackBodies": ["Pelvis", "Head", "L_Hand", "R_Hand"]}, {"obs_v": 9}, {"obs_v": 8},
                                  {"obs_v": 2}, {"self_obs_v": 3}, {"self_obs_v": 2, "past_track_steps": 3}, {"has_upright_start": False},
                                  {"enableEarlyTermination": False}])
def test_humanoid_im_observation_variants_step(dev, over):
    n, horizon = 66, 6
    env, _ = configs.make_env(n, horizon, dev, seed=11, reference="motion_lib", env_overrides=over)
    task = env.task
    obs = env.reset()
    assert obs.shape == (n, task.num_obs) and torch.isfinite(obs).all()
    for _ in range(4):
        prev = task.sim.rigid_body_state.clone()
        obs, rew, done, info = env.step(torch.zeros(n, 69, device=dev))
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    rb = task.sim.rigid_body_state.cpu()
    bp, br, bv, ba = E.split_rb(rb)
    up = task._has_upright_start
    if task.self_obs_v == 1:
        _cmp(obs[:, :358], E.self_obs_smpl_max_general(bp, br, bv, ba, True, True, up))
    elif task.self_obs_v == 3:
        assert task._self_obs_size == 358 + 12
        _cmp(obs[:, :358], E.self_obs_smpl_max_general(bp, br, bv, ba, True, True, up))
        assert (obs[:, 358:370] == 0).all()
    else:
        h = task._rb_hist.cpu()
        assert h.shape[1] == 4 and torch.equal(h[:, -1], rb) and torch.equal(h[:, -2], prev.cpu())
        _cmp(obs[:, :task._self_obs_size], E.self_obs_smpl_max_v2(h[..., 0:3].contiguous(), h[..., 3:7].contiguous(), h[..., 7:10].contiguous(),
                                                                  h[..., 10:13].contiguous(), True, True, up))
    if over.get("enableEarlyTermination", True) is False:
        assert info["terminate"].sum() == 0


def test_speed_z_policy_trains_end_to_end(dev):
    agent, _ = configs.make_agent("speed_z_small", device=str(dev), seed=5)
    assert agent.actions_num == 32 and agent.model.in_dim == 361                  # latent action, 358 + 3 observation (env_pulse_amp.yaml)
    before = agent.model.flat.clone()
    for _ in range(2):
        info = agent.train_epoch()
    assert all(torch.isfinite(torch.as_tensor(x)).all() for x in info["actor_loss"]) and not torch.equal(before, agent.model.flat)
    eb = agent.experience_buffer
    assert eb.tensor_dict["actions"].shape[-1] == 32 and torch.isfinite(eb.tensor_dict["rewards"]).all()


@pytest.mark.parametrize("name", ["HumanoidSpeed", "HumanoidStrike"])
def test_power_usage_reward_steps_match_oracle(dev, name):
    """[r6] power_usage_reward (humanoid_speed.py:225-238, humanoid_strike.py:186-198): the left / right power-balance term on top of the
    kernel's reward, the accumulator carried across steps (cleared on reset by the speed task only), reward_raw's extra column (speed only)."""
    n, frames = 96, 12
    sim = HT.SyntheticTaskSim(n, frames, dev, seed=8)
    speed = name == "HumanoidSpeed"
    task = HT.TASKS[name]({"env": {"power_reward": speed, "power_usage_reward": True, "power_usage_coefficient": 0.004, "episode_length": 7}}, sim, device=dev)
    task.reset()
    left, right = TO.side_dof_indexes(syn.SMPL_BODY_NAMES[1:], lower_only=not speed)
    acc = torch.zeros(n, 2)
    saw_reset = 0
    for step in range(12):
        prev_root = sim.rigid_body_state[:, 0, 0:3].clone().cpu()
        prog = task.progress_buf.clone().cpu() + 1
        task.step(torch.zeros(n, 69, device=dev))
        rb = sim.rigid_body_state.cpu()
        root = rb[:, 0]
        if speed:
            rew = TO.speed_reward(root[:, 0:3], prev_root, root[:, 3:7], task._tar_speed.cpu(), task.dt)
            pw = -0.0005 * (sim.dof_force * sim.dof_vel).abs().sum(-1).cpu()
            pw[prog <= 3] = 0
            rew = rew + pw
        else:
            ts = task._target_states.cpu()
            rew = TO.strike_reward(ts[:, 0:3], ts[:, 3:7], root, prev_root, rb[:, 23, 7:10], task.dt, 1.4)
        pur = TO.power_usage_reward(sim.dof_force.cpu(), sim.dof_vel.cpu(), acc, prog, left, right, 0.004)
        _cmp(task.rew_buf, rew + pur)
        _cmp(task.power_acc, acc, atol=1e-3)
        raw = task.extras["reward_raw"]
        assert raw.shape[1] == (3 if speed else 1)
        if speed:
            _cmp(raw[:, 2], pur)
        done = task.reset_buf.bool()
        saw_reset += int(done.sum())
        task.reset_masked(done)
        if speed:
            acc[done.cpu()] = 0
    assert saw_reset > 0 and (pur != 0).any()
    # the other task classes do not read the key (the reference's HumanoidReach has no such block): accepted and without effect
    reach = HT.HumanoidReach({"env": {"power_usage_reward": True}}, HT.SyntheticTaskSim(8, 4, dev, seed=1), device=dev)
    assert reach.power_usage_reward is False
