"""The frozen PHC teacher inside ``env.step`` (PULSE distillation target), on the gfx950 GEMM plans.

Mirrors HumanoidImDistill.step (phc/env/tasks/humanoid_im_distill.py:143-231, ``has_pnn`` branch):

    full_obs  = clamp((obs - running_mean) / sqrt(running_var + 1e-5), +-5)          teacher's own statistics (:165-183)
    x_k       = pnn.actors[k](full_obs),  k < num_prim                                 PNN without lateral links (pnn.py:84-131)
    weights   = composer(full_obs)                                                     MLP WITH a trailing activation (network_loader.py:40-42)
    gt_action = sum_k weights[:, k, None] * x_k                                        (:195-198)

and the loaders ``load_pnn`` / ``load_mcp_mlp`` (phc/learning/network_loader.py:11-73): layer sizes are read off the checkpoint
tensors, keys ``a2c_network.pnn.actors.<k>.<2i>.{weight,bias}`` and ``a2c_network.composer.<2i>.{weight,bias}``.
All num_prim + 1 MLPs read the same normalised observation buffer; the primitives' outputs land side by side in one
(N, num_prim * 72) buffer, so the mixture is one broadcast-multiply-reduce.
Only the no-lateral PNN (every released PHC checkpoint: ``has_lateral: False``) is supported.
"""
import torch

from .. import kernels as K
from .._lib import ACT_RELU, ACT_SILU
from .graph import MlpGraph, ParamBook, r4

_ACTS = {"relu": ACT_RELU, "silu": ACT_SILU}


def _layer_sizes(model, prefix):
    """[(out, in), ...] of ``prefix.<0,2,4,...>.weight`` in Sequential order."""
    sizes, i = [], 0
    while f"{prefix}.{2 * i}.weight" in model:
        sizes.append(tuple(model[f"{prefix}.{2 * i}.weight"].shape))
        i += 1
    if not sizes:
        raise KeyError(f"no {prefix}.*.weight tensors in the checkpoint")
    return sizes


class PnnTeacher:
    def __init__(self, pnn_checkpoint, composer_checkpoint, num_prim, num_envs, activation="silu", has_lateral=False, device="cuda:0"):
        if has_lateral:
            raise NotImplementedError("PNN lateral connections (pnn.py:90-123) are not used by the released teachers")
        if activation not in _ACTS:
            raise NotImplementedError(f"teacher activation {activation!r}: relu / silu are built")
        pm, cm = pnn_checkpoint["model"], composer_checkpoint["model"]
        act = _ACTS[activation]
        self.device, self.n, self.num_prim = torch.device(device), num_envs, num_prim
        self.book = ParamBook(self.device, split_k=1)
        g = self.g = MlpGraph(self.book, num_envs)
        first = _layer_sizes(pm, "a2c_network.pnn.actors.0")
        self.in_dim, self.num_actions = first[0][1], first[-1][0]
        self.a_pitch = r4(self.num_actions)
        self.x = g.buffer("x", self.in_dim)
        g.buffer("acts", num_prim * self.a_pitch)
        self._lins = []
        for k in range(num_prim):
            sizes = _layer_sizes(pm, f"a2c_network.pnn.actors.{k}")
            units = [s[0] for s in sizes[:-1]]
            self._lins += g.mlp(self.book, f"a2c_network.pnn.actors.{k}", "x", self.in_dim, units, act, [f"p{k}h{i}" for i in range(len(units))],
                                final_linear=sizes[-1][0], final_dst="acts", final_dst_col=k * self.a_pitch)
        csizes = _layer_sizes(cm, "a2c_network.composer")
        cunits = [s[0] for s in csizes]
        if cunits[-1] != num_prim:
            raise ValueError(f"composer emits {cunits[-1]} weights for {num_prim} primitives")
        self._clins = g.mlp(self.book, "a2c_network.composer", "x", self.in_dim, cunits, act, [f"ch{i}" for i in range(len(cunits) - 1)] + ["w"])
        self.book.finalize(trainable=False)
        for lin in self._lins:
            self.book.set(lin.w.name, pm[lin.w.name])
            self.book.set(lin.b.name, pm[lin.b.name])
        for lin in self._clins:
            self.book.set(lin.w.name, cm[lin.w.name])
            self.book.set(lin.b.name, cm[lin.b.name])
        rms = pnn_checkpoint["running_mean_std"]
        self.running_mean = rms["running_mean"].to(self.device, torch.float64).contiguous()
        self.running_var = rms["running_var"].to(self.device, torch.float64).contiguous()
        self._plan = g.forward_plan(store_pre=False)
        self.gt_action = torch.zeros(num_envs, self.num_actions, device=self.device)

    def parameters_count(self):
        return self.book.n_flat

    def forward(self, obs_store):
        """obs_store: the env's (N, pitch) observation rows (first in_dim columns used) -> gt_action (N, 69)."""
        n = self.n
        K.rms_normalize(obs_store, self.running_mean, self.running_var, rows=n, cols=self.in_dim, x_stride=obs_store.stride(0), y=self.x,
                        y_stride=self.x.stride(0), y_cols=self.x.shape[1], clip=5.0)
        self._plan.run()
        acts = self.g.act_bufs["acts"].view(n, self.num_prim, self.a_pitch)[:, :, :self.num_actions]
        w = self.g.act_bufs["w"][:, :self.num_prim]
        torch.sum(w[:, :, None] * acts, dim=1, out=self.gt_action)
        return self.gt_action
