"""Reference-motion library resident in HBM (mirrors the query surface of phc/utils/motion_lib_base.py).

What the reference does per env step (humanoid_im.py:708-735, 853-861, 950-964): twice, gather two frames from six flat
tables (gts, grs, lrs, gvs, gavs, dvs; motion_lib_base.py:297-304) for every env, lerp / slerp them, turn the local
rotations into exp-map dof positions and add the per-env offset -- ~60 small launches and twelve scattered gathers.

Here the tables are packed ONCE at load time into one record per frame (include/pulse_hip.h section 2b:
[gts | grs | lrs | gvs | gavs | dvs | pad], 480 floats = 1920 B for SMPL) and ``get_motion_state`` is one launch of
``pulse_motion_state`` that reads two contiguous records per query.  Clip loading from AMASS pickles / SkeletonMotion
construction (motion_lib_base.py:172-285, motion_lib_smpl.py) is CPU-side data preparation and stays out of scope;
``MotionLib`` takes the finished tables (``from_tables``) -- synthetic clips here (pulse_amd/synthetic.py).

Same names and return keys as MotionLibBase: ``get_motion_state``, ``get_root_pos_smpl``, ``get_motion_length``,
``get_motion_num_steps``, ``sample_motions``, ``sample_time``, ``sample_time_interval``, ``num_motions``,
``get_total_length``; attributes ``_motion_lengths``, ``_motion_fps``, ``_motion_dt``, ``_motion_num_frames``,
``length_starts``, ``_sampling_prob``.
"""
import ctypes

import torch

from .. import _lib
from .._lib import MotionStateArgs


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _round4(n):
    return (n + 3) // 4 * 4


class MotionLib:
    FIELDS = ("gts", "grs", "lrs", "gvs", "gavs", "dvs")

    def __init__(self, tables, device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError("MotionLib lives in HBM (pulse_amd has no CPU path)")
        self._device = dev
        gts = tables["gts"]
        total, j = gts.shape[0], gts.shape[1]
        self.num_bodies = j
        self.num_dof = (j - 1) * 3
        widths = {"gts": j * 3, "grs": j * 4, "lrs": j * 4, "gvs": j * 3, "gavs": j * 3, "dvs": (j - 1) * 3}
        # quaternion fields on 16-B boundaries: order [grs | lrs | gts | gvs | gavs | dvs] keeps them aligned for any J
        order = ("grs", "lrs", "gts", "gvs", "gavs", "dvs")
        self.offsets, off = {}, 0
        for k in order:
            self.offsets[k] = off
            off += widths[k]
        self.frame_stride = _round4(off)
        frames = torch.zeros(total, self.frame_stride, dtype=torch.float32, device=dev)
        for k in order:
            t = tables[k]
            if t.shape[0] != total or t.dtype != torch.float32:
                raise ValueError(f"table {k}: expected ({total}, ...) float32")
            frames[:, self.offsets[k]:self.offsets[k] + widths[k]] = t.reshape(total, -1).to(dev)
        self.frames = frames
        self._motion_lengths = tables["motion_lengths"].to(dev, torch.float32).contiguous()
        self._motion_fps = tables["motion_fps"].to(dev, torch.float32).contiguous()
        self._motion_dt = tables["motion_dt"].to(dev, torch.float32).contiguous()
        self._motion_num_frames = tables["motion_num_frames"].to(dev, torch.int64).contiguous()
        lengths_shifted = self._motion_num_frames.roll(1)                      # motion_lib_base.py:311-314
        lengths_shifted[0] = 0
        self.length_starts = lengths_shifted.cumsum(0).contiguous()
        self._num_motions = self._motion_lengths.shape[0]
        self.motion_ids = torch.arange(self._num_motions, dtype=torch.long, device=dev)
        self._sampling_prob = torch.ones(self._num_motions, device=dev) / self._num_motions      # :205, uniform until re-weighted
        self._sampling_batch_prob = self._sampling_prob
        self._lengths_host = None
        # PMCP bookkeeping over the data set's clips (motion_lib_base.py:196-206): here every clip of the tables is resident, so the
        # "unique motions" of the data set and the loaded batch coincide
        self._num_unique_motions = self._num_motions
        keys = tables.get("motion_data_keys")
        self._motion_data_keys = list(keys) if keys is not None else [str(i) for i in range(self._num_motions)]
        self._termination_history = torch.zeros(self._num_unique_motions, device=dev)

    @classmethod
    def from_tables(cls, tables, device="cuda:0"):
        return cls(tables, device)

    # ---- table views (reference attribute names), strided views into the packed records
    def _field(self, k, inner):
        j = self.num_bodies if k != "dvs" else self.num_bodies - 1
        return self.frames[:, self.offsets[k]:self.offsets[k] + j * inner].view(self.frames.shape[0], j, inner)

    gts = property(lambda self: self._field("gts", 3))
    grs = property(lambda self: self._field("grs", 4))
    lrs = property(lambda self: self._field("lrs", 4))
    gvs = property(lambda self: self._field("gvs", 3))
    gavs = property(lambda self: self._field("gavs", 3))
    dvs = property(lambda self: self._field("dvs", 3))

    # ---- bookkeeping queries (motion_lib_base.py:325-432)
    def num_motions(self):
        return self._num_motions

    def get_total_length(self):
        if self._lengths_host is None:
            self._lengths_host = float(self._motion_lengths.sum().item())
        return self._lengths_host

    def get_motion_length(self, motion_ids=None):
        return self._motion_lengths if motion_ids is None else self._motion_lengths[motion_ids]

    def get_motion_num_steps(self, motion_ids=None):
        if motion_ids is None:
            return (self._motion_num_frames * 30 / self._motion_fps).int()
        return (self._motion_num_frames[motion_ids] * 30 / self._motion_fps).int()

    # ---- PMCP sampling weights (motion_lib_base.py:348-393; IMAmpAgent.update_training_data, im_amp.py:126-132).  In the reference
    # _sampling_prob steers which clips load_motions makes resident (:212-222) and the batch distribution follows from it; with every
    # clip resident, load_motions here only refreshes the batch distribution from the weights.
    def update_hard_sampling_weight(self, failed_keys):
        """auto_pmcp: train only on the sequences that failed evaluation (:348-360)."""
        if len(failed_keys) > 0:
            indexes = [self._motion_data_keys.index(k) for k in failed_keys]
            self._sampling_prob[:] = 0
            self._sampling_prob[indexes] = 1 / len(indexes)
        else:
            self._sampling_prob = torch.ones(self._num_unique_motions, device=self._device) / self._num_unique_motions

    def update_soft_sampling_weight(self, failed_keys):
        """auto_pmcp_soft: sampling weight proportional to how often a sequence failed evaluation (:362-376)."""
        if len(failed_keys) > 0:
            indexes = [self._motion_data_keys.index(k) for k in failed_keys]
            self._termination_history[indexes] += 1
            self.update_sampling_prob(self._termination_history)
        else:
            self._sampling_prob = torch.ones(self._num_unique_motions, device=self._device) / self._num_unique_motions

    def update_sampling_prob(self, termination_history):
        """:378-384."""
        termination_history = torch.as_tensor(termination_history, dtype=torch.float32, device=self._device)
        if len(termination_history) == len(self._termination_history) and termination_history.sum() > 0:
            self._sampling_prob[:] = termination_history / termination_history.sum()
            self._termination_history = termination_history
            return True
        return False

    def load_motions(self):
        """The part of MotionLibBase.load_motions (:172-285) that survives when every clip is resident: the batch distribution is the
        data-set distribution restricted to the loaded clips and renormalised (:221-222)."""
        self._sampling_batch_prob = self._sampling_prob / self._sampling_prob.sum()

    def sample_motions(self, n, generator=None):
        return torch.multinomial(self._sampling_batch_prob, num_samples=n, replacement=True, generator=generator).to(self._device)

    def sample_time(self, motion_ids, truncate_time=None, generator=None):
        phase = torch.rand(motion_ids.shape, device=self._device, generator=generator)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len = motion_len - truncate_time
        return phase * motion_len

    def sample_time_interval(self, motion_ids, truncate_time=None, generator=None):
        phase = torch.rand(motion_ids.shape, device=self._device, generator=generator)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len = motion_len - truncate_time
        curr_fps = 1 / 30
        return ((phase * motion_len) / curr_fps).long() * curr_fps

    # ---- the hot query
    def launch_signature(self):
        """Identity of the device tables a cached launch points at (ops._launch_sig)."""
        return (id(self), self.frames.data_ptr(), tuple(self.frames.shape), self._motion_lengths.data_ptr(), self._motion_dt.data_ptr(),
                self._motion_num_frames.data_ptr(), self.length_starts.data_ptr(), self._num_motions)

    def fill_tables(self, t):
        """Fill a pulse_motion_tables struct (by reference) with this library's device pointers."""
        t.frames, t.frame_stride, t.total_frames, t.num_bodies = self.frames.data_ptr(), self.frame_stride, self.frames.shape[0], self.num_bodies
        o = self.offsets
        t.off_gts, t.off_grs, t.off_lrs, t.off_gvs, t.off_gavs, t.off_dvs = o["gts"], o["grs"], o["lrs"], o["gvs"], o["gavs"], o["dvs"]
        t.motion_lengths, t.motion_dt = self._motion_lengths.data_ptr(), self._motion_dt.data_ptr()
        t.motion_num_frames, t.length_starts, t.num_motions = self._motion_num_frames.data_ptr(), self.length_starts.data_ptr(), self._num_motions

    def query(self, motion_ids, motion_times=None, offset=None, *, progress=None, step_shift=0, dt=0.0, start_times=None,
              start_offsets=None, time_steps=1, traj_dt=0.0, out=None, root_only=False, with_frames=False, with_records=False,
              reset=None, fields=None):
        """One launch.  Times either given (``motion_times``) or built in-kernel from the episode clock
        ((progress + step_shift) * dt + start_times + start_offsets).  ``out``: dict of preallocated outputs to reuse."""
        lib = _lib.load()
        dev = self._device
        ids = motion_ids
        if ids.dtype != torch.int64 or not ids.is_cuda:
            raise TypeError("motion_ids: int64 device tensor expected")
        ids = ids.contiguous()
        ne = ids.numel()                                   # per-env arrays
        n = ne if motion_times is not None else ne * int(time_steps)
        j, nd = self.num_bodies, self.num_dof
        keep = [ids]

        def f32(t, name, shape):
            if t is None:
                return None
            if t.dtype != torch.float32 or not t.is_cuda:
                raise TypeError(f"{name}: float32 device tensor expected")
            t = t.contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
            keep.append(t)
            return t

        a = MotionStateArgs()
        self.fill_tables(a.tab)
        a.n, a.motion_ids = n, ids.data_ptr()
        if reset is not None:
            # reset mode (include/pulse_hip.h 2b): masked envs get a new start time and their reference state in one launch
            m = reset["mask"]
            m = (m.view(torch.uint8) if m.dtype == torch.bool else m).contiguous()
            keep.append(m)
            a.reset_mask, a.step_shift, a.dt = m.data_ptr(), int(step_shift), float(dt)
            ph = f32(reset.get("phase"), "reset.phase", (ne,))
            a.reset_phase = ph.data_ptr() if ph is not None else None
            a.reset_time_interval = int(bool(reset.get("time_interval", False)))
            for field, key, dt_ in (("reset_start_times", "start_times", torch.float32), ("reset_progress", "progress", torch.int64),
                                    ("reset_clear0", "clear0", torch.int64), ("reset_clear1", "clear1", torch.int64),
                                    ("reset_clear2", "clear2", torch.int64), ("reset_start_offsets", "zero_start_offsets", torch.float32),
                                    ("reset_global_offset", "zero_global_offset", torch.float32)):
                t_ = reset.get(key)
                if t_ is not None:
                    if t_.dtype != dt_ or not t_.is_contiguous() or t_.numel() != (3 * ne if key == "zero_global_offset" else ne):
                        raise TypeError(f"reset.{key}: contiguous {dt_} tensor of {ne} elements expected")
                    setattr(a, field, t_.data_ptr())
            so = f32(start_offsets, "start_offsets", (ne,))
            a.start_offsets = so.data_ptr() if so is not None else None
        elif motion_times is not None:
            a.motion_times = f32(motion_times, "motion_times", (n,)).data_ptr()
        else:
            if progress is None or progress.dtype != torch.int64:
                raise TypeError("query needs motion_times or an int64 progress tensor")
            progress = progress.contiguous()
            keep.append(progress)
            a.progress, a.step_shift, a.dt = progress.data_ptr(), int(step_shift), float(dt)
            a.time_steps, a.traj_dt = int(time_steps), float(traj_dt)
            st, so = f32(start_times, "start_times", (ne,)), f32(start_offsets, "start_offsets", (ne,))
            a.start_times = st.data_ptr() if st is not None else None
            a.start_offsets = so.data_ptr() if so is not None else None
        off = f32(offset, "offset", (ne, 3))
        a.offset = off.data_ptr() if off is not None else None
        res = {} if out is None else out
        e = lambda key, *shape, dtype=torch.float32: res[key] if key in res else res.setdefault(key, torch.empty(*shape, dtype=dtype, device=dev))
        if root_only:
            a.root_only = 1
            a.root_pos = e("root_pos", n, 3).data_ptr()
        else:
            shapes = {"rg_pos": (n, j, 3), "rb_rot": (n, j, 4), "body_vel": (n, j, 3), "body_ang_vel": (n, j, 3), "dof_pos": (n, nd),
                      "dof_vel": (n, nd), "rb_records": (n, j, 13)}
            want = fields if fields is not None else ("rg_pos", "rb_rot", "body_vel", "body_ang_vel", "dof_pos", "dof_vel") + (("rb_records",) if with_records else ())
            for k in want:
                t_ = e(k, *shapes[k])
                if tuple(t_.shape) != shapes[k] or t_.dtype != torch.float32 or not t_.is_contiguous():
                    raise ValueError(f"out[{k}]: contiguous float32 {shapes[k]} expected")
                setattr(a, k, t_.data_ptr())
                if k == "rb_records":
                    a.rb_query_stride = t_.stride(0)
        if with_frames:
            a.frame_idx0, a.frame_idx1 = e("frame_idx0", n, dtype=torch.int64).data_ptr(), e("frame_idx1", n, dtype=torch.int64).data_ptr()
            a.blend = e("blend", n).data_ptr()
        _lib.check(lib.pulse_motion_state(ctypes.byref(a), _stream()), "pulse_motion_state")
        return res

    def get_motion_state(self, motion_ids, motion_times, offset=None, out=None):
        """MotionLibBase.get_motion_state (motion_lib_base.py:434-517).  Root entries are views of body 0 (the reference
        clones them); motion_aa / motion_bodies / motion_limb_weights (SMPL shape metadata) are not carried."""
        res = self.query(motion_ids, motion_times, offset, out=out)
        res["root_pos"], res["root_rot"] = res["rg_pos"][:, 0], res["rb_rot"][:, 0]
        res["root_vel"], res["root_ang_vel"] = res["body_vel"][:, 0], res["body_ang_vel"][:, 0]
        return res

    def get_root_pos_smpl(self, motion_ids, motion_times):
        return self.query(motion_ids, motion_times, root_only=True)
