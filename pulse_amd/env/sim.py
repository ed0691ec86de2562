"""Stand-ins for the two things the hot path reads but does not own: the simulator's state
tensors (Isaac Gym / PhysX, closed source, OUT OF SCOPE) and the reference-motion frames.

``RecordedRollout`` holds pre-recorded synthetic frames generated once on the CPU
(pulse_amd/synthetic.py) and resident in HBM -- "identical pre-recorded rollout buffers" that both
the CPU oracle and the HIP path consume.  ``RecordedSim`` exposes them the way Isaac Gym exposes
its tensors to phc/env/tasks/humanoid.py:175-243 (a persistent (N, bodies, 13) rigid-body buffer
that "physics" overwrites every step, dof force / velocity tensors, a PD-target setter), and
``RecordedMotion`` plays the role of ``MotionLibBase.get_motion_state``
(phc/utils/motion_lib_base.py:434-517) for the two evaluations per step (time t and t+1).
"""
import torch

from .. import kernels as K
from .. import synthetic as syn


class RecordedRollout:
    """frames[f] for f in 0..num_frames-1: everything one env step reads."""

    def __init__(self, num_envs, num_frames, seed=1234, rank=0, done_rate=0.02):
        g = syn.make_generator(seed, rank)
        self.num_envs, self.num_frames = num_envs, num_frames
        n = num_envs
        keys = ("rb", "reset_rb", "dof_force", "dof_vel", "gt_action", "dof_pos")
        self.data = {k: [] for k in keys}
        self.ref_now = {k: [] for k in ("pos", "rot", "vel", "ang")}
        self.ref_next = {k: [] for k in ("pos", "rot", "vel", "ang")}
        self.ref_next_reset = {k: [] for k in ("pos", "rot", "vel", "ang")}
        for _ in range(num_frames):
            d = syn.env_step_inputs(g, n, edge_cases=False)
            self.data["rb"].append(d["rb"])
            self.data["dof_force"].append(d["dof_force"])
            self.data["dof_vel"].append(d["dof_vel"])
            for k in self.ref_now:
                self.ref_now[k].append(d["ref_now"][k])
                self.ref_next[k].append(d["ref_next"][k])
            # stand-in for the frozen PHC teacher's action on this frame (humanoid_im_distill.py:143-231; needs released
            # checkpoints, so the distillation target is synthetic)
            self.data["gt_action"].append((0.4 * torch.randn(n, syn.NUM_DOF, generator=g)).clamp(-1, 1))
            self.data["dof_pos"].append(0.5 * torch.randn(n, syn.NUM_DOF, generator=g))       # exp-map joint angles
            rrb = syn.rigid_body_state(g, n)
            self.data["reset_rb"].append(rrb)
            rr = syn.reference_frame(g, rrb)
            for k in self.ref_next_reset:
                self.ref_next_reset[k].append(rr[k])
        self.data = {k: torch.stack(v) for k, v in self.data.items()}
        self.ref_now = {k: torch.stack(v) for k, v in self.ref_now.items()}
        self.ref_next = {k: torch.stack(v) for k, v in self.ref_next.items()}
        self.ref_next_reset = {k: torch.stack(v) for k, v in self.ref_next_reset.items()}
        # episode clock: motion length / start time so that ~done_rate of the envs time out per step
        self.dt = 2.0 / 60.0                                                    # humanoid.py:122, controlFrequencyInv 2
        self.motion_lengths = 1.0 + 2.0 * (1.0 / max(done_rate, 1e-6)) * self.dt * torch.rand(n, generator=g)
        self.motion_start_times = torch.zeros(n)
        self.init_progress = torch.randint(0, 40, (n,), generator=g, dtype=torch.int64)
        # some envs drift far from their reference -> early termination
        far = torch.rand(num_frames, n, generator=g) < done_rate / 2
        self.ref_now["pos"][:, :, 13, :] += far[..., None].float() * 1.0

    def to(self, device):
        self.data = {k: v.to(device) for k, v in self.data.items()}
        for d in (self.ref_now, self.ref_next, self.ref_next_reset):
            for k in d:
                d[k] = d[k].to(device)
        self.motion_lengths = self.motion_lengths.to(device)
        self.motion_start_times = self.motion_start_times.to(device)
        self.init_progress = self.init_progress.to(device)
        return self


class RecordedSim:
    """Isaac-Gym-shaped tensor API over a RecordedRollout (device resident)."""

    def __init__(self, rollout):
        self.rollout = rollout
        dev = rollout.data["rb"].device
        n = rollout.num_envs
        self.num_envs = n
        self.frame = 0
        self.rigid_body_state = rollout.data["rb"][0].clone()               # the persistent "gym tensor"
        self.dof_force = rollout.data["dof_force"][0].clone()
        self.dof_vel = rollout.data["dof_vel"][0].clone()
        self.dof_pos = rollout.data["dof_pos"][0].clone()
        self.pd_targets = torch.zeros(n, syn.NUM_DOF, device=dev)

    @property
    def gt_action(self):
        return self.rollout.data["gt_action"][self.frame]

    def rewind(self):
        self.frame = 0
        self._load(0)

    def _load(self, f):
        r = self.rollout.data
        self.rigid_body_state.copy_(r["rb"][f])
        self.dof_force.copy_(r["dof_force"][f])
        self.dof_vel.copy_(r["dof_vel"][f])
        self.dof_pos.copy_(r["dof_pos"][f])

    def set_dof_position_target_tensor(self, pd_tar):
        # the recorded "physics" ignores the targets; they are kept so the write happens as in
        # phc/env/tasks/humanoid.py:1246-1247
        self.pd_targets = pd_tar

    def simulate_and_refresh(self):
        """control_freq_inv x gym.simulate + refresh_*_tensor (humanoid.py:1282-1297, humanoid_amp.py:599-620)."""
        self.frame = (self.frame + 1) % self.rollout.num_frames
        self._load(self.frame)

    def set_env_states_masked(self, mask):
        """Reference-state init of the masked envs (humanoid_amp.py:447-488): their records are
        replaced by this frame's recorded reset state."""
        rr = self.rollout.data["reset_rb"][self.frame]
        torch.where(mask[:, None, None], rr, self.rigid_body_state, out=self.rigid_body_state)


class RecordedMotion:
    """get_motion_state stand-in: the frames recorded for the current sim frame."""

    def __init__(self, rollout, sim):
        self.rollout, self.sim = rollout, sim
        self._motion_lengths = rollout.motion_lengths
        self._demo_gen = torch.Generator(device=rollout.data["rb"].device)
        self._demo_gen.manual_seed(4321)

    def sample_demo_states(self, count):
        """Synthetic stand-in for the motion-library states behind fetch_amp_obs_demo (humanoid_amp.py:215-284):
        ``count`` random reference poses -> (rb records (count, 24, 13), dof_pos, dof_vel), drawn on the device."""
        dev, g = self.rollout.data["rb"].device, self._demo_gen
        rb = torch.randn(count, syn.NUM_BODIES, syn.RB_WIDTH, device=dev, generator=g)
        rb[..., 0:3] *= 0.3
        rb[:, :, 2] += 0.9
        q = rb[..., 3:7]
        rb[..., 3:7] = q / q.norm(dim=-1, keepdim=True)
        dof_pos = 0.4 * torch.randn(count, syn.NUM_DOF, device=dev, generator=g)
        dof_vel = 0.8 * torch.randn(count, syn.NUM_DOF, device=dev, generator=g)
        return rb, dof_pos, dof_vel

    def now(self):
        f = self.sim.frame
        return {k: v[f] for k, v in self.rollout.ref_now.items()}

    def next(self):
        f = self.sim.frame
        return {k: v[f] for k, v in self.rollout.ref_next.items()}

    def next_after_reset(self):
        f = self.sim.frame
        return {k: v[f] for k, v in self.rollout.ref_next_reset.items()}


class KinematicSim:
    """Physics stand-in for the motion-library path: the simulated humanoid TRACKS the reference motion with recorded
    perturbations.  ``simulate_and_refresh`` sets the rigid-body state to the reference state the env asked it to
    track (the t+1 reference of the previous step) plus this frame's noise record; reference-state init
    (humanoid_amp.py:447-488, humanoid_im.py:966-986) writes the motion state unperturbed.  Same tensor surface as
    RecordedSim.  The noise bank is generated on the CPU (pulse_amd/synthetic.py generator) so a CPU twin can replay it."""

    def __init__(self, num_envs, bank_frames, device, seed=1234, rank=0, drift_rate=0.01):
        g = syn.make_generator(seed + 17, rank)
        n, j, f = num_envs, syn.NUM_BODIES, bank_frames
        self.num_envs, self.bank_frames = n, f
        noise = torch.randn(f, n, j, syn.RB_WIDTH, generator=g)
        noise[..., 0:3] *= 0.03
        noise[..., 3:7] *= 0.04
        noise[..., 7:10] *= 0.15
        noise[..., 10:13] *= 0.3
        far = torch.rand(f, n, generator=g) < drift_rate          # some envs lose their reference -> early termination
        noise[:, :, 13, 0:3] += far[..., None].float() * 1.0
        self.bank = {"rb": noise, "dof_force": 50.0 * torch.randn(f, n, syn.NUM_DOF, generator=g),
                     "dof_pos": 0.02 * torch.randn(f, n, syn.NUM_DOF, generator=g), "dof_vel": 0.1 * torch.randn(f, n, syn.NUM_DOF, generator=g),
                     "gt_action": (0.4 * torch.randn(f, n, syn.NUM_DOF, generator=g)).clamp(-1, 1)}
        self.bank = {k: v.to(device) for k, v in self.bank.items()}
        self.frame = 0
        self.rigid_body_state = torch.zeros(n, j, syn.RB_WIDTH, device=device)
        self.rigid_body_state[..., 6] = 1.0
        self.dof_force = self.bank["dof_force"][0].clone()
        self.dof_vel = torch.zeros(n, syn.NUM_DOF, device=device)
        self.dof_pos = torch.zeros(n, syn.NUM_DOF, device=device)
        self.pd_targets = torch.zeros(n, syn.NUM_DOF, device=device)
        self._target = None

    @property
    def gt_action(self):
        return self.bank["gt_action"][self.frame]

    def set_dof_position_target_tensor(self, pd_tar):
        self.pd_targets = pd_tar

    def track(self, state):
        """state: a MotionLib query result with rb_records / dof_pos / dof_vel (the reference at the NEXT control step)."""
        self._target = state

    def simulate_and_refresh(self):
        self.frame = (self.frame + 1) % self.bank_frames
        f, t, b = self.frame, self._target, self.bank
        K.kinematic_sim_step(t["rb_records"], b["rb"][f], self.rigid_body_state, t["dof_pos"], b["dof_pos"][f], self.dof_pos,
                             t["dof_vel"], b["dof_vel"][f], self.dof_vel, b["dof_force"][f], self.dof_force)      # one launch

    def set_env_states_masked(self, mask, state):
        torch.where(mask[:, None, None], state["rb_records"], self.rigid_body_state, out=self.rigid_body_state)
        torch.where(mask[:, None], state["dof_pos"], self.dof_pos, out=self.dof_pos)
        torch.where(mask[:, None], state["dof_vel"], self.dof_vel, out=self.dof_vel)


class PdSim(KinematicSim):
    """Physics stand-in whose state DEPENDS ON THE ACTION (pulse_pd_sim_step): per-joint error states under PD control towards
    ``sag + action_scale * action`` with a recorded disturbance; bodies are the tracked reference displaced by those errors.  Used
    by the return-parity experiment (tools/return_parity.py) -- the bench keeps KinematicSim.  The CPU twin is
    oracle/pd_sim_oracle.py; both draw the disturbance bank with the same generator recipe."""

    def __init__(self, num_envs, bank_frames, device, seed=1234, rank=0):
        super().__init__(num_envs, bank_frames, device, seed=seed, rank=rank)
        g = syn.make_generator(seed + 23, rank)
        self.bank["acc"] = (syn.PD_SIM["noise_acc"] * torch.randn(bank_frames, num_envs, syn.NUM_DOF, generator=g)).to(device)
        sag, u = syn.pd_sim_tables()
        self._sag, self._lever_dir = sag.to(device), u.to(device).contiguous()
        self.err = torch.zeros(num_envs, syn.NUM_DOF, device=device)
        self.err_vel = torch.zeros(num_envs, syn.NUM_DOF, device=device)
        self.dt = 2.0 / 60.0
        self._reset_mask = torch.zeros(num_envs, dtype=torch.uint8, device=device)

    def simulate_and_refresh(self):
        from .. import _lib
        self.frame = (self.frame + 1) % self.bank_frames
        t, c = self._target, syn.PD_SIM
        a = _lib.PdSimArgs()
        a.num_envs, a.num_bodies = self.num_envs, syn.NUM_BODIES
        a.target_rb, a.target_dof_pos, a.target_dof_vel = t["rb_records"].data_ptr(), t["dof_pos"].data_ptr(), t["dof_vel"].data_ptr()
        act = self.pd_targets.contiguous()
        a.action, a.noise_acc = act.data_ptr(), self.bank["acc"][self.frame].data_ptr()
        a.sag, a.lever_dir = self._sag.data_ptr(), self._lever_dir.data_ptr()
        a.kp, a.kd, a.dt, a.action_scale, a.lever, a.substeps = c["kp"], c["kd"], self.dt, c["action_scale"], c["lever"], c["substeps"]
        a.err, a.err_vel = self.err.data_ptr(), self.err_vel.data_ptr()
        a.rb, a.dof_pos, a.dof_vel, a.dof_force = (self.rigid_body_state.data_ptr(), self.dof_pos.data_ptr(), self.dof_vel.data_ptr(),
                                                   self.dof_force.data_ptr())
        import ctypes
        _lib.check(_lib.load().pulse_pd_sim_step(ctypes.byref(a), torch.cuda.current_stream().cuda_stream), "pulse_pd_sim_step")

    def on_reset(self, mask):
        """Reference-state init of the masked envs (HumanoidIm.reset_masked wrote their state): the error states restart at zero."""
        keep = (~mask).to(self.err.dtype)[:, None]
        self.err.mul_(keep)
        self.err_vel.mul_(keep)

    def set_env_states_masked(self, mask, state):
        super().set_env_states_masked(mask, state)
        self.on_reset(mask)
