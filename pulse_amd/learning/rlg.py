"""The slice of rl_games 1.1.4 the PULSE agents lean on, rebuilt for device-resident, sync-free use.

rl_games is a third-party dependency of the reference (requirement.txt:27) that is absent here;
the semantics below follow the reference's call sites (phc/learning/common_agent.py,
amp_agent.py, amp_datasets.py) and the public 1.1.4 behaviour summarised in SURVEY.md Appendix B.

  ExperienceBuffer  a2c_common.A2CBase.init_tensors / experience.ExperienceBuffer
  AMPDataset        phc/learning/amp_datasets.py:36-100 over rl_games datasets.PPODataset
  AverageMeter      torch_ext.AverageMeter (windowed mean of finished-episode stats)
  IdentityScheduler / AdaptiveScheduler   rl_games schedulers
  DefaultRewardsShaper

MI355X-first choices:
  * rollout tensors are stored ENV-MAJOR, physically (N, T, .), and exposed through the reference's
    (T, N, .) indexing as strided views.  ``swap_and_flatten01`` -- a transposing copy of every
    tensor in the reference (~1 GB of traffic at 4096 x 32 x 934) -- becomes a free reshape;
  * observation rows keep the GEMM-ready pitch (934 -> 960) all the way into the dataset;
  * nothing here reads a device value back to the host.
"""
import torch


def swap_and_flatten01(arr):
    """a2c_common.swap_and_flatten01: (T, N, ...) -> (N*T, ...) with row index env*T + t."""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


class ExperienceBuffer:
    def __init__(self, num_actors, horizon_length, obs_dim, obs_pitch, actions_num, action_pitch, device):
        self.num_actors, self.horizon_length = num_actors, horizon_length
        self.device = torch.device(device)
        self.obs_base_shape = (horizon_length, num_actors)
        n, t = num_actors, horizon_length
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=self.device)
        # physical storage, env-major
        self.phys = {
            "obses": z(n, t, obs_pitch), "rewards": z(n, t, 1), "values": z(n, t, 1), "neglogpacs": z(n, t),
            "dones": z(n, t, dtype=torch.uint8), "actions": z(n, t, action_pitch), "mus": z(n, t, action_pitch),
            "sigmas": z(n, t, action_pitch),
        }
        self.widths = {"obses": obs_dim, "actions": actions_num, "mus": actions_num, "sigmas": actions_num}
        self.tensor_dict = {}
        for k in self.phys:
            self._expose(k)

    def _expose(self, k):
        v = self.phys[k].transpose(0, 1)                  # (T, N, pitch) view
        w = self.widths.get(k)
        self.tensor_dict[k] = v[..., :w] if w is not None else v

    def add(self, name, like=None, width=None, pitch=None, dtype=torch.float32):
        n, t = self.num_actors, self.horizon_length
        if like is not None:
            self.phys[name] = torch.zeros_like(self.phys[like])
            if like in self.widths:
                self.widths[name] = self.widths[like]
        else:
            self.phys[name] = torch.zeros(n, t, pitch or width, dtype=dtype, device=self.device)
            if pitch and pitch != width:
                self.widths[name] = width
        self._expose(name)

    def slot(self, name, index):
        """(N, pitch) strided view of time step ``index`` in physical storage."""
        return self.phys[name][:, index]

    def update_data(self, name, index, val):
        dst = self.tensor_dict[name][index]
        if isinstance(val, torch.Tensor) and val.data_ptr() == dst.data_ptr() and val.stride() == dst.stride():
            return                                      # the producer already wrote in place
        dst.copy_(val)

    def get_transformed_list(self, transform_op, tensor_list):
        return {k: transform_op(self.tensor_dict[k]) for k in tensor_list if k in self.tensor_dict}

    def flat(self, name):
        """(N*T, pitch) view of the physical storage (row = env*T + t), pitch preserved."""
        p = self.phys[name]
        return p.reshape(p.shape[0] * p.shape[1], *p.shape[2:])


class AMPDataset:
    """Random-permutation minibatcher (amp_datasets.py:81-100).  ``__getitem__`` returns the index
    slice plus references to the un-gathered tensors: the gather is fused into the consuming
    kernels (normaliser, PPO loss).  ``gather(i)`` materialises the reference-style dict."""

    NUM_STAGING = 16

    def __init__(self, batch_size, minibatch_size, is_discrete, is_rnn, device, seq_len, generator=None, permutation_device="cpu"):
        # 'cpu': drawn like the reference (torch.randperm on the host, amp_datasets.py:7) and uploaded -- a shared seed reproduces the
        # oracle's minibatches; 'cuda': drawn on the device -- no host shuffle (7 ms for 131072 rows), no staging copy, nothing the
        # launch thread can stall on between mini-epochs
        self.permutation_device = permutation_device
        self._dev_gen = None
        self.is_rnn = bool(is_rnn)          # use_seq_rl: minibatches are whole env sequences (amp_datasets.py:36-79)
        self.horizon_length, self.num_envs = 1, batch_size
        self.batch_size, self.minibatch_size = batch_size, minibatch_size
        self.device = torch.device(device)
        self.length = batch_size // minibatch_size
        self.generator = generator
        self.special_names = ["rnn_states"]
        self.values_dict = None
        self._pinned = None
        self._pin_i = 0
        self._idx_buf = self._randperm()

    def _randperm(self, n=None):
        """Drawn on the CPU like the reference (torch.randperm(self.batch_size), amp_datasets.py:7) so a shared
        seed reproduces the oracle's minibatches, then uploaded through a rotating PINNED staging buffer with a
        non-blocking copy: a pageable .to(device) would stall the host until the GPU drains its queue."""
        n = self.batch_size if n is None else n
        if self.device.type != "cuda":
            return torch.randperm(n, generator=self.generator)
        if self.permutation_device == "cuda":
            if self._dev_gen is None:
                self._dev_gen = torch.Generator(device=self.device)
                self._dev_gen.manual_seed(self.generator.initial_seed() if self.generator is not None else 0)
            return torch.randperm(n, device=self.device, generator=self._dev_gen)
        if self._pinned is None:
            # enough staging buffers that a buffer is only re-used after a whole epoch (whose end synchronises the device):
            # the guard below then never has to BLOCK -- a blocking event wait with GPU work in flight can cost ~60 ms here
            self._pinned = [torch.empty(self.batch_size, dtype=torch.int64).pin_memory() for _ in range(self.NUM_STAGING)]
            self._pin_done = [None] * self.NUM_STAGING
        i = self._pin_i % self.NUM_STAGING
        self._pin_i += 1
        buf = self._pinned[i][:n]
        if self._pin_done[i] is not None and not self._pin_done[i].query():
            self._pin_done[i].synchronize()      # the host may run several mini-epochs ahead of the GPU: never overwrite a
                                                 # staging buffer whose upload has not executed yet (rare with NUM_STAGING buffers)
        # Fisher-Yates straight into the pinned buffer would be ~25 ms: pinned host memory is uncached for the CPU on ROCm
        # and the shuffle is random access.  Shuffle in ordinary memory, then stream the 1 MB into the staging buffer.
        buf.copy_(torch.randperm(n, generator=self.generator))
        dev = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pin_done[i] = ev
        return dev

    def set_permutation(self, perm):
        self._idx_buf = perm.to(self.device, torch.int64)

    def update_values_dict(self, values_dict, rnn_format=False, horizon_length=1, num_envs=1):
        self.values_dict = values_dict
        self.horizon_length, self.num_envs = horizon_length, num_envs
        if rnn_format and self.is_rnn:
            # rows are already env-major (row = env * T + t), so "view(num_envs, T, -1)" is implicit; only the
            # permutation changes: it shuffles ENVS (amp_datasets.py:47)
            self._perm_n = num_envs
            self._idx_buf = self._randperm(num_envs)
            self._t_range = torch.arange(horizon_length, device=self.device)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        if self.is_rnn:
            return self._get_item_rnn(idx)
        start, end = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        out = {"idx": self._idx_buf[start:end], "dataset": self.values_dict}
        if end >= self.batch_size:
            self._shuffle_idx_buf()
        return out

    def _get_item_rnn(self, idx):
        """amp_datasets.py:54-79: step_size = minibatch // T env sequences, rows of each env kept in time order."""
        t = self.horizon_length
        step = self.minibatch_size // t
        envs = self._idx_buf[idx * step:(idx + 1) * step]
        rows = (envs[:, None] * t + self._t_range[None, :]).reshape(-1)
        out = {"idx": rows, "dataset": self.values_dict, "num_seqs": step}
        # amp_datasets.py:75-76 compares ``end`` -- an ENV index here -- with batch_size (= T * num_envs): that never fires, so the
        # reference re-uses the env permutation drawn in update_values_dict for every mini-epoch of the epoch.  Mirrored.
        if (idx + 1) * step >= self.batch_size:
            self._idx_buf = self._randperm(self._perm_n)
        return out

    def gather(self, idx):
        start, end = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        sample_idx = self._idx_buf[start:end]
        return {k: v[sample_idx] for k, v in self.values_dict.items() if k not in self.special_names and v is not None}

    def _shuffle_idx_buf(self):
        self._idx_buf = self._randperm()


class AverageMeter:
    """torch_ext.AverageMeter (rl_games) with device-resident state ``[mean, current_size]`` so the
    rollout's per-step update runs inside pulse_rollout_record; ``update_masked`` is the same update for other callers."""

    def __init__(self, in_shape, max_size, device):
        if tuple(in_shape) != (1,):
            raise NotImplementedError("scalar meters (value_size 1) are what the hot path tracks")
        self.max_size = max_size
        self.state = torch.zeros(2, dtype=torch.float32, device=device)
        self.mean = self.state[0:1]
        self.current_size = self.state[1]

    def update_masked(self, values, mask):
        """values (N, ...) / mask (N,) bool: same as update(values[mask]) without materialising it."""
        m = mask.to(values.dtype)
        size = m.sum()
        w = m.view(-1, *([1] * (values.dim() - 1)))
        new_mean = (values * w).sum(0) / torch.clamp(size, min=1.0)
        size_c = torch.clamp(size, 0, self.max_size)
        old_size = torch.minimum(self.max_size - size_c, self.current_size)
        size_sum = old_size + size_c
        upd = (self.mean * old_size + new_mean * size_c) / torch.clamp(size_sum, min=1.0)
        has = size > 0
        self.mean.copy_(torch.where(has, upd, self.mean))
        self.current_size.copy_(torch.where(has, size_sum, self.current_size))

    def clear(self):
        self.current_size.zero_()
        self.mean.zero_()

    def get_mean(self):
        return self.mean.squeeze(0).cpu().numpy()


class IdentityScheduler:
    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        return current_lr, entropy_coef


class AdaptiveScheduler:
    def __init__(self, kl_threshold=0.008):
        self.min_lr, self.max_lr, self.kl_threshold = 1e-6, 1e-2, kl_threshold

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        lr = current_lr
        if kl_dist > (2.0 * self.kl_threshold):
            lr = max(current_lr / 1.5, self.min_lr)
        if kl_dist < (0.5 * self.kl_threshold):
            lr = min(current_lr * 1.5, self.max_lr)
        return lr, entropy_coef


class DefaultRewardsShaper:
    def __init__(self, scale_value=1, shift_value=0, min_val=-float("inf"), max_val=float("inf")):
        self.scale_value, self.shift_value, self.min_val, self.max_val = scale_value, shift_value, min_val, max_val
        self.identity = scale_value == 1 and shift_value == 0 and min_val == -float("inf") and max_val == float("inf")

    def __call__(self, reward):
        if self.identity:
            return reward
        reward = (reward + self.shift_value) * self.scale_value
        return torch.clamp(reward, self.min_val, self.max_val)
