"""Data parallelism over environments: one process per GPU, RCCL over xGMI via torch.distributed.

Replaces rl_games' HorovodWrapper as the reference uses it:
  setup_algo        common_agent.py:112-113   broadcast parameters + optimiser state from rank 0
  synchronize       common_agent.py:465-471, amp_agent.py:736-742   gradient all-reduce (average)
                    after backward, before clip / Adam
  average_value     common_agent.py:224-247   scalar mean of the KL for the LR scheduler
  sync_stats        common_agent.py:126-127   per-epoch averaging of the running-stat buffers

The path shards by environment (no cross-env term in obs / reward / reset / GAE; advantage
normalisation and running statistics are per rank in the reference), so the only exchange step per
optimiser step is ONE all-reduce of the flat gradient buffer (12 MB for the [1024, 512] actor +
critic).  xGMI is point-to-point (7 links x ~153 GB/s): a single flat bucket keeps the ring at
its per-link bound with one launch instead of one per parameter tensor.  The 1/world_size
averaging is folded into the gradient-slab reduce that precedes the all-reduce.

Backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
import os

import torch
import torch.distributed as dist


class DistContext:
    def __init__(self, enabled=None, backend=None):
        ws_env = int(os.environ.get("WORLD_SIZE", "1"))
        self.enabled = (ws_env > 1) if enabled is None else enabled
        self.rank, self.world_size, self.local_rank = 0, 1, 0
        self._own_group = False
        self._stat = {"calls": 0, "bytes": 0, "seconds": 0.0}
        self._pending = []
        self.time_exposure, self._exposure_events = False, []
        if not self.enabled:
            return
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.world_size = ws_env
        if not dist.is_initialized():
            if backend is None:
                # PULSE_DIST_BACKEND=gloo lets a 1-GPU box exercise the multi-rank control flow (ranks share the device)
                backend = os.environ.get("PULSE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
            self._own_group = True

    # ---- collectives -----------------------------------------------------------------------------
    def broadcast_(self, t, src=0):
        if self.enabled:
            dist.broadcast(t, src=src)
        return t

    def all_reduce_sum_(self, t):
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def average_value(self, t, name=None):
        """hvd.allreduce(mean) of a scalar / small tensor."""
        if not self.enabled:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / self.world_size

    def barrier(self):
        if self.enabled:
            dist.barrier()

    def max_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = torch.device("cuda", self.local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    # ---- agent-level protocol ----------------------------------------------------------------------
    def setup_algo(self, flat_params, extra_tensors=()):
        """Broadcast the flat parameter buffer (+ optimiser state / constants) from rank 0."""
        self.broadcast_(flat_params)
        for t in extra_tensors:
            self.broadcast_(t)

    def sync_gradients(self, flat_grad, async_op=False):
        """The explicit `optimizer.synchronize()` of the reference: SUM all-reduce of the flat gradient
        (each rank pre-scaled its gradient by 1/world_size in the slab reduce).  ``async_op``: the collective is enqueued behind
        the work already on the current stream and runs beside what is launched next; ``wait_gradients`` joins it."""
        import time
        t0 = time.perf_counter()
        if async_op and self.enabled:
            self._pending.append(dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True))
            out = flat_grad
        else:
            out = self.all_reduce_sum_(flat_grad)
        self._stat["calls"] += 1
        self._stat["bytes"] += flat_grad.numel() * flat_grad.element_size()
        self._stat["seconds"] += time.perf_counter() - t0            # host enqueue time (the collective itself is asynchronous)
        return out

    def wait_gradients(self):
        """Make the current stream wait for every gradient bucket in flight.  With ``time_exposure`` on (bench.py --gpus N) the wait is
        bracketed by two events on the compute stream: their distance is the part of the all-reduce the overlap did NOT hide (device
        time, read back once after the timed region)."""
        if not self._pending:
            return
        ev = None
        if self.time_exposure and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._pending:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._exposure_events.append(ev)
        self._pending = []

    def exposed_wait_ms(self):
        """Sum of the device-side waits recorded by wait_gradients (call after a synchronize)."""
        ms = sum(a.elapsed_time(b) for a, b in self._exposure_events)
        n = len(self._exposure_events)
        self._exposure_events = []
        return ms, n

    def backend_world_size(self):
        """The rank count as the BACKEND sees it (an RCCL communicator that came up with fewer ranks than WORLD_SIZE would show here)."""
        return dist.get_world_size() if (self.enabled and dist.is_initialized()) else 1

    def stats(self):
        d = dict(self._stat)
        d["backend"] = dist.get_backend() if (self.enabled and dist.is_initialized()) else "none"
        return d

    def sync_stats(self, stat_modules, curr_frames):
        """Average running mean / var / count across ranks; sum the frame counter."""
        if not self.enabled:
            return curr_frames
        for m in stat_modules:
            if m is None:
                continue
            for buf in (m.running_mean, m.running_var, m.count):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                buf /= self.world_size
            if hasattr(m, "_count_host"):
                # every rank adds the same batch sizes, so the averaged count equals the local one;
                # keep the host mirror consistent without a device read
                pass
        dev = stat_modules[0].running_mean.device if stat_modules and stat_modules[0] is not None else "cpu"
        t = torch.tensor([float(curr_frames)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def shutdown(self):
        if self._own_group and dist.is_initialized():
            dist.destroy_process_group()
