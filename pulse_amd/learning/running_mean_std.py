"""RunningMeanStd on the GPU (mirrors phc/utils/running_mean_std.py:9-109).

Same contract as the reference module: fp64 ``running_mean`` / ``running_var`` / ``count`` buffers
(count starts at 1), ``forward`` normalises with the CURRENT statistics and clamps to +-5, and --
in training mode and unless frozen -- merges the batch moments (unbiased variance) AFTER the
output has been produced (:98-107).  The arithmetic runs in the HIP kernels
``pulse_rms_normalize`` / ``pulse_rms_update``; this class only owns the buffers.

Differences that are deliberate (MI355X-first):
  * ``forward`` can gather rows (``row_idx``) and write into a caller-supplied, GEMM-ready
    pitched buffer with zero-filled padding columns, so "minibatch gather -> normalise -> first
    Linear" needs no intermediate copies;
  * ``count`` is mirrored on the host (it only ever grows by the batch size) so no kernel has to
    read a scalar back.
Only the non-per-channel, 1-D ``insize`` form used by the hot path is implemented.
"""
import torch

from .. import kernels as K


class RunningMeanStd:
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False, device="cuda:0"):
        if per_channel or norm_only:
            raise NotImplementedError("per_channel / norm_only normalisers are not on the PULSE hot path")
        if isinstance(insize, int):
            insize = (insize,)
        if len(insize) != 1:
            raise NotImplementedError("only 1-D observation shapes are supported")
        self.insize = tuple(insize)
        self.mean_size = insize[0]
        self.epsilon = epsilon
        self.device = torch.device(device)
        self.running_mean = torch.zeros(self.mean_size, dtype=torch.float64, device=self.device)
        self.running_var = torch.ones(self.mean_size, dtype=torch.float64, device=self.device)
        self.count = torch.ones((), dtype=torch.float64, device=self.device)
        self._count_host = 1.0
        self.training = True
        self.forzen = False          # (sic) attribute name kept from the reference
        self._partials = None

    # ---- nn.Module-like surface -------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def freeze(self):
        self.forzen = True

    def unfreeze(self):
        self.forzen = False

    def to(self, device):
        return self

    def state_dict(self):
        return {"running_mean": self.running_mean.clone(), "running_var": self.running_var.clone(), "count": self.count.clone()}

    def load_state_dict(self, sd):
        self.running_mean.copy_(sd["running_mean"].to(self.device, torch.float64))
        self.running_var.copy_(sd["running_var"].to(self.device, torch.float64))
        self.count.copy_(sd["count"].to(self.device, torch.float64))
        self._count_host = float(sd["count"])          # one host read at load time only

    def clone_frozen(self):
        """copy.deepcopy(self) + freeze(), as AMPAgent.pre_epoch does (amp_agent.py:578-579)."""
        c = RunningMeanStd(self.insize, self.epsilon, device=self.device)
        c.running_mean.copy_(self.running_mean)
        c.running_var.copy_(self.running_var)
        c.count.copy_(self.count)
        c._count_host = self._count_host
        c.training = self.training
        c.forzen = True
        return c

    # ---- forward ------------------------------------------------------------------------------
    def _moment_buffer(self, rows):
        nblk = max(1, min(512, rows // 16))
        if self._partials is None or self._partials.shape[0] != nblk:
            self._partials = torch.zeros(nblk, 2, self.mean_size, dtype=torch.float64, device=self.device)
        return self._partials

    def forward(self, input, unnorm=False, *, row_idx=None, out=None, out_cols=None, update=None, norm_with=None, planes=None, raw_out=None):
        """input: (rows, >=mean_size) float32 with unit inner stride.  Returns ``out`` (allocated
        (rows, mean_size) when not given).  ``update`` overrides the training/frozen rule."""
        x = input
        if x.dim() == 1:
            x = x.unsqueeze(-1)
        rows = x.shape[0] if row_idx is None else row_idx.numel()
        f = self.mean_size
        if out is None:
            out = torch.empty(rows, f if out_cols is None else out_cols, dtype=torch.float32, device=x.device)
        do_update = (self.training and not self.forzen) if update is None else update
        do_update = do_update and not unnorm
        part = self._moment_buffer(rows) if do_update else None
        src = self if norm_with is None else norm_with       # AMPAgent._preproc_obs(use_temp): output from the frozen copy,
        K.rms_normalize(x, src.running_mean, src.running_var, rows=rows, cols=f, x_stride=x.stride(0), y=out,  # update of the live stats
                        y_stride=out.stride(0), y_cols=out.shape[1] if out_cols is None else out_cols, row_idx=row_idx,
                        eps=self.epsilon, unnorm=unnorm, moment_partials=part,
                        num_blocks=None if part is None else part.shape[0], planes=planes, raw_out=raw_out)
        if do_update:
            K.rms_update(self.running_mean, self.running_var, self.count, part, f, self._count_host, rows)
            self._count_host += rows
        return out if input.dim() > 1 else out.squeeze(-1)

    __call__ = forward
