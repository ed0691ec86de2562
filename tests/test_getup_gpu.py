"""GPU: HumanoidImGetup (phc/env/tasks/humanoid_im_getup.py:42-210) -- the three-way reset split (recovery episode / fall-state start /
reference-state init), the recovery grace period of _compute_reset, the getup schedule."""
import pytest
import torch

from pulse_amd import configs, synthetic as syn
from pulse_amd.env.humanoid_im_getup import HumanoidImGetup
from pulse_amd.env.motion_lib import MotionLib
from pulse_amd.env.sim import KinematicSim

pytestmark = pytest.mark.gpu


def make(dev, n=96, with_tables=False, **env):
    cfg = dict(configs.ENV_IM)
    cfg.update(env)
    tables = syn.synthetic_motion_library(syn.make_generator(11, 0), n)
    lib = MotionLib.from_tables(tables, dev)
    sim = KinematicSim(n, 40, dev, seed=3)
    task = HumanoidImGetup({"env": cfg}, sim, lib, device=dev)
    return (task, tables) if with_tables else task


def test_three_way_reset_split_and_recovery_grace(dev):
    n = 96
    task = make(dev, n, recoveryEpisodeProb=1.0, fallInitProb=1.0, recoverySteps=5)
    task.reset()                                              # initial reset: nothing was terminated -> every env starts from ITS fall state
    assert torch.equal(task.sim.rigid_body_state, task._fall_state["rb_records"][task._last_fall_perm])      # a permutation: no state shared
    assert torch.equal(torch.sort(task._last_fall_perm).values, torch.arange(n, device=dev))
    assert (task._recovery_counter == 5).all() and (task.progress_buf == 0).all()
    assert (task.sim.rigid_body_state[..., 7:13] == 0).all() and (task.sim.rigid_body_state[..., 2].min(dim=1).values - 0.05).abs().max() < 1e-5
    # during the grace period nothing resets or terminates and progress stands still (:203-210)
    for k in range(4):
        task.step(torch.zeros(n, 69, device=dev))
        assert (task.reset_buf == 0).all() and (task._terminate_buf == 0).all() and (task.progress_buf == 0).all()
        assert (task._recovery_counter == 4 - k).all()
    # grace over: the ordinary termination logic is back
    for _ in range(30):
        task.step(torch.zeros(n, 69, device=dev))
    assert (task.progress_buf > 0).any()
    # terminated envs become recovery episodes (prob 1): their state is untouched, counter re-armed; timed-out envs are not terminated -> fall start
    task._terminate_buf[:] = 0
    task._terminate_buf[:10] = 1
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    mask[:20] = True
    before = task.sim.rigid_body_state.clone()
    task.reset_masked(mask)
    assert torch.equal(task.sim.rigid_body_state[:10], before[:10])                            # recovery: state kept
    assert torch.equal(task.sim.rigid_body_state[10:20], task._fall_state["rb_records"][task._last_fall_perm][10:20])   # fall start
    assert torch.equal(task.sim.rigid_body_state[20:], before[20:])
    assert (task._recovery_counter[:20] == 5).all() and (task.progress_buf[:20] == 0).all() and (task.reset_buf[:20] == 0).all()
    assert torch.isfinite(task.obs_buf).all()


def test_getup_schedule_and_normal_init(dev):
    n = 64
    task = make(dev, n, recoveryEpisodeProb=0.5, fallInitProb=0.3, recoverySteps=7, getup_schedule=True)
    task.update_getup_schedule(5, getup_udpate_epoch=10)
    assert (task._recovery_episode_prob, task._fall_init_prob) == (0.0, 1.0)
    task.update_getup_schedule(11, getup_udpate_epoch=10)
    assert (task._recovery_episode_prob, task._fall_init_prob) == (0.5, 0.3)
    # no recovery, no fall starts: the reset is the ordinary reference-state init and the counter is cleared
    task._recovery_episode_prob, task._fall_init_prob = 0.0, 0.0
    task._recovery_counter[:] = 3
    task.reset()
    ref = task._motion_lib.query(task._sampled_motion_ids, task._motion_start_times, task._global_offset, with_records=True)
    assert torch.equal(task.sim.rigid_body_state, ref["rb_records"]) and (task._recovery_counter == 0).all()
    # mixed probabilities: roughly fallInitProb of the non-terminated resets start fallen
    task._fall_init_prob = 0.3
    task.reset()
    frac = float(task._reset_fall_mask.float().mean())
    assert 0.1 < frac < 0.55, frac


@pytest.mark.parametrize("cycle", [False, True])
def test_recovering_env_observes_the_frozen_clock_frame(dev, cycle):
    """humanoid_im_getup.py:203-210 + humanoid.py:1325-1328: _compute_reset takes the recovering envs' progress back BEFORE
    _compute_observations, so their task observation targets the frame at (frozen progress + 1), not one frame further."""
    from oracle import env_oracle as E
    from oracle.motion_oracle import OracleMotionLib
    n = 64
    task, tables = make(dev, n, with_tables=True, recoveryEpisodeProb=1.0, fallInitProb=1.0, recoverySteps=6, cycle_motion=cycle)
    olib = OracleMotionLib(tables)                                   # CPU restatement of MotionLibBase.get_motion_state (pinned to the reference)
    task.reset()
    for _ in range(3):
        task.step(torch.zeros(n, 69, device=dev))
    assert (task.progress_buf == 0).all() and (task._recovery_counter == 3).all()

    def expected(shift):
        """The observation row the REFERENCE builds for this state, computed on the CPU by the oracle (motion query + self / task
        observation functions) from copies of the env's state: the reference frame at (progress + shift) * dt."""
        t = (task.progress_buf.cpu() + shift).float() * task.dt + task._motion_start_times.cpu() + task._motion_start_times_offset.cpu()
        ref = olib.get_motion_state(task._sampled_motion_ids.cpu(), t, task._global_offset.cpu())
        bp, br, bv, ba = E.split_rb(task.sim.rigid_body_state.cpu())
        tb = [int(i) for i in task._track_bodies_id.cpu().tolist()] if isinstance(task._track_bodies_id, torch.Tensor) else list(task._track_bodies_id)
        self_obs = E.self_obs_smpl_max(bp, br, bv, ba)
        assert task.obs_v == 6
        task_obs = E.im_obs_v6(bp[:, 0], br[:, 0], bp[:, tb], br[:, tb], bv[:, tb], ba[:, tb], ref["rg_pos"][:, tb], ref["rb_rot"][:, tb],
                               ref["body_vel"][:, tb], ref["body_ang_vel"][:, tb], 1)
        return torch.cat([self_obs, task_obs], dim=-1)
    w = task.num_obs
    want, ahead = expected(1), expected(2)
    got = task.obs_buf[:, :w].cpu()
    assert (got - want).abs().max().item() <= 1e-5, (got - want).abs().max().item()
    assert (got - ahead).abs().max().item() > 1e-3                   # one frame further is a different row
    # once the grace period is over the clock advances again and the observation follows it
    for _ in range(4):
        task.step(torch.zeros(n, 69, device=dev))
    moving = (task.progress_buf > 0).cpu()
    assert moving.any()
    assert (task.obs_buf[:, :w].cpu()[moving] - expected(1)[moving]).abs().max().item() <= 1e-5
