"""HumanoidImGetup on MI355X: recovery episodes and fall-state episode initialisation for the imitation env.

Mirrors phc/env/tasks/humanoid_im_getup.py:42-210 (the env PULSE is distilled in: HumanoidImDistillGetup, env_im_vae.yaml:2):
  update_getup_schedule   :67-74   before ``getup_udpate_epoch`` every reset is a fall start and no recovery episodes are granted
  _generate_fall_states   :82-127  a bank of fallen states (the reference drops every humanoid with random actions for 150 physics
                                   steps; physics is out of scope, so the bank comes from ``fall_state_source`` -- by default the
                                   motion's first pose tumbled about a random axis and laid on the ground, velocities zeroed :110-121)
  _reset_actors           :137-165 three-way split of the envs being reset: terminated envs become RECOVERY episodes with probability
                                   recoveryEpisodeProb (state untouched, recoverySteps of grace); of the rest a fraction fallInitProb starts
                                   from a fall state (same grace); the others get the ordinary reference-state init
  _compute_reset          :203-210 envs in recovery neither reset nor terminate and their progress does not advance
  _init_amp_obs           :189-196 fall starts fill the AMP history with the current frame (_init_amp_obs_default)
Everything is masked tensor code (no ``nonzero()`` read-backs).  Fall starts take a random state of the bank through a fresh
permutation per reset (the reference draws random FREE state ids, :171-183: no state is shared by two envs either way).
"""
import torch

from .. import ops
from .humanoid_im import HumanoidIm


class HumanoidImGetup(HumanoidIm):
    def __init__(self, cfg, sim, motion_lib, device="cuda:0", fall_state_source=None):
        env = cfg.get("env", cfg)
        self._recovery_episode_prob_tgt = self._recovery_episode_prob = float(env.get("recoveryEpisodeProb", 0.5))
        self._recovery_steps_tgt = self._recovery_steps = int(env.get("recoverySteps", 90))
        self._fall_init_prob_tgt = self._fall_init_prob = float(env.get("fallInitProb", 0.3))
        self.getup_udpate_epoch = int(env.get("getup_udpate_epoch", 10000))
        self.getup_schedule = bool(env.get("getup_schedule", False))
        super().__init__(cfg, sim, motion_lib, device=device)
        if not self._use_motion_lib:
            raise NotImplementedError("HumanoidImGetup needs the MotionLib reference source (reference-state init of the non-fall envs)")
        n, dev = self.num_envs, self.device
        self._recovery_counter = torch.zeros(n, dtype=torch.int32, device=dev)
        self._reset_fall_mask = torch.zeros(n, dtype=torch.bool, device=dev)
        self._getup_gen = torch.Generator(device=dev)
        self._getup_gen.manual_seed(int(env.get("getup_seed", 4711)))
        self._fall_source = fall_state_source
        self._generate_fall_states()

    # ------------------------------------------------------------------ :67-74
    def update_getup_schedule(self, epoch_num, getup_udpate_epoch=5000):
        if epoch_num > getup_udpate_epoch:
            self._recovery_episode_prob, self._fall_init_prob = self._recovery_episode_prob_tgt, self._fall_init_prob_tgt
        else:
            self._recovery_episode_prob, self._fall_init_prob = 0.0, 1.0

    # ------------------------------------------------------------------ :82-127
    def _generate_fall_states(self):
        if self._fall_source is not None:
            st = self._fall_source(self)
        else:
            lib, ids = self._motion_lib, self._sampled_motion_ids
            ref = lib.query(ids, torch.zeros(self.num_envs, device=self.device), self._global_offset, with_records=True)
            rb = ref["rb_records"].clone()
            q = torch.randn(self.num_envs, 4, device=self.device, generator=self._getup_gen)     # random root rotation (:90-91)
            q = q / q.norm(dim=-1, keepdim=True)
            j = rb.shape[1]
            qe = q[:, None].expand(-1, j, -1).reshape(-1, 4)
            root = rb[:, :1, 0:3]
            rel = ops.my_quat_rotate(qe, (rb[..., 0:3] - root).reshape(-1, 3)).view(-1, j, 3)
            rb[..., 0:3] = rel + root
            rb[..., 2] = rb[..., 2] - rb[..., 2].min(dim=1, keepdim=True).values + 0.05           # lying on the ground
            rb[..., 3:7] = ops.quat_mul(qe, rb[..., 3:7].reshape(-1, 4).contiguous()).view(-1, j, 4)
            rb[..., 7:13] = 0                                                                     # :110-111, :121
            st = {"rb_records": rb.contiguous(), "dof_pos": ref["dof_pos"].clone(), "dof_vel": torch.zeros_like(ref["dof_vel"])}
        self._fall_state = st

    def resample_motions(self):
        self._generate_fall_states()
        self.reset()

    # ------------------------------------------------------------------ step phases
    def pre_physics_step(self, actions):
        super().pre_physics_step(actions)
        self._recovery_counter.sub_(1).clamp_(min=0)                                              # _update_recovery_count (:198-201)

    def _recovery_counter_for_step(self):
        # _compute_reset (:203-210) runs INSIDE the fused step: envs in recovery neither reset nor terminate, their progress does not
        # advance and -- because the reference decrements progress before _compute_observations (humanoid.py:1325-1328) -- their next
        # observation targets the frame of the frozen clock
        return self._recovery_counter

    # ------------------------------------------------------------------ :137-196 for the masked envs
    def reset_masked(self, mask):
        n = self.num_envs
        u = torch.rand(2, n, device=self.device, generator=self._getup_gen)
        recovery = mask & (u[0] < self._recovery_episode_prob) & (self._terminate_buf == 1)
        fall = mask & ~recovery & (u[1] < self._fall_init_prob)
        normal = mask & ~recovery & ~fall
        self._reset_fall_mask = fall
        # ordinary reference-state init for the normal envs (clears their flags / clock); recovery and fall envs keep or replace
        # their state below, but every env being reset gets progress 0 and cleared flags (_reset_env_tensors)
        super().reset_masked(normal)
        both = recovery | fall
        if hasattr(self.sim, "set_env_states_masked"):
            # a random FREE fall state per env (_reset_fall_episode, :171-183): a permutation of the bank, so no state is shared
            perm = torch.randperm(n, device=self.device, generator=self._getup_gen)
            self._last_fall_perm = perm
            # gathered into persistent buffers (index_select with out=): no permuted copy of the bank is allocated per step
            if getattr(self, "_fall_pick", None) is None:
                self._fall_pick = {k: torch.empty_like(v) for k, v in self._fall_state.items()}
            for k, v in self._fall_state.items():
                torch.index_select(v, 0, perm, out=self._fall_pick[k])
            self.sim.set_env_states_masked(fall, self._fall_pick)
        if self.self_obs_v == 2:
            self._init_tensor_history(fall)           # a fall start's history is its own state repeated (humanoid.py:1301-1306)
        keep = (~both)
        self.progress_buf.mul_(keep)
        self.reset_buf.mul_(keep)
        self._terminate_buf.mul_(keep)
        self._recovery_counter.copy_(torch.where(both, torch.full_like(self._recovery_counter, self._recovery_steps),
                                                 torch.where(normal, torch.zeros_like(self._recovery_counter), self._recovery_counter)))
        self._compute_observations(env_mask=both)
        if self._enable_amp_obs:
            self._compute_amp_observations(env_mask=both)
            # fall starts: history = current frame (_init_amp_obs_default, :189-196); recovery envs keep their history
            s = self._num_amp_obs_steps - 1
            hist = self._curr_amp_obs_buf.unsqueeze(1).expand(-1, s, -1)
            self._hist_amp_obs_buf.copy_(torch.where(fall[:, None, None], hist, self._hist_amp_obs_buf))
