"""The frozen PHC teacher inside ``env.step`` (PULSE distillation target), on the gfx950 GEMM plans.

Mirrors HumanoidImDistill.step (phc/env/tasks/humanoid_im_distill.py:143-231, ``has_pnn`` branch):

    full_obs  = clamp((obs - running_mean) / sqrt(running_var + 1e-5), +-5)          teacher's own statistics (:165-183)
    x_k       = pnn.actors[k](full_obs),  k < num_prim                                 PNN forward, with or without lateral links (pnn.py:84-131)
    weights   = composer(full_obs)                                                     MLP WITH a trailing activation (network_loader.py:40-42)
    gt_action = sum_k weights[:, k, None] * x_k                                        (:195-198)

and the loaders ``load_pnn`` / ``load_mcp_mlp`` (phc/learning/network_loader.py:11-73): layer sizes are read off the checkpoint
tensors, keys ``a2c_network.pnn.actors.<k>.<2i>.{weight,bias}`` and ``a2c_network.composer.<2i>.{weight,bias}``.
All num_prim + 1 MLPs read the same normalised observation buffer; the primitives' outputs land side by side in one
(N, num_prim * 72) buffer, so the mixture is one broadcast-multiply-reduce.
``has_lateral: True`` (pnn.py:24-37, 90-123; the released PHC checkpoints use False): column c's second layer adds
``sum_{j<c} u[c-1][j][0](h1_j)`` to its pre-activation before the activation (the action-space transfer ``u[..][1]`` is disabled in
the reference, :104).  Here the first-layer outputs of all columns sit side by side in ONE buffer and column c's second layer is a
single GEMM over the contiguous range [h1_0 .. h1_c] against the row-concatenated weights [u[c-1][0] | .. | u[c-1][c-1] | W2_c]:
no extra launches, no adds.  Like the reference (:96) this supports two hidden layers per column.
"""
import torch

from .. import kernels as K
from .._lib import ACT_RELU, ACT_SILU
from .graph import Linear, MlpGraph, ParamBook, r4

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
        self._lins, self._lat = [], []
        self.has_lateral = bool(has_lateral)
        if self.has_lateral:
            if len(first) != 3:
                raise NotImplementedError("lateral PNN columns have exactly two hidden layers (pnn.py:96)")
            u0, u1 = first[0][0], first[1][0]
            if u0 % 4:
                raise ValueError("first hidden width must be a multiple of 4 floats")
            g.buffer("h1all", num_prim * u0)
            for k in range(num_prim):                                   # every column's first layer lands in its slice of h1all
                lin = Linear(self.book, f"a2c_network.pnn.actors.{k}.0", self.in_dim, u0, act)
                g.linear(lin, "x", "h1all", dst_col=k * u0)
                self._lins.append(lin)
            for k in range(num_prim):
                lat = Linear(self.book, f"pnn_lateral.{k}", (k + 1) * u0, u1, act)          # [u[k-1][0..k-1][0] | actors[k][2]]
                g.buffer(f"p{k}h2", u1)
                g.linear(lat, "h1all", f"p{k}h2")
                self._lat.append((lat, k, u0))
                lin = Linear(self.book, f"a2c_network.pnn.actors.{k}.4", u1, self.num_actions)
                g.linear(lin, f"p{k}h2", "acts", dst_col=k * self.a_pitch)
                self._lins.append(lin)
        else:
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
        for lat, k, u0 in self._lat:
            cols = [pm[f"a2c_network.pnn.u.{k - 1}.{j}.0.weight"] for j in range(k)] + [pm[f"a2c_network.pnn.actors.{k}.2.weight"]]
            self.book.set(lat.w.name, torch.cat([c.to(torch.float32) for c in cols], dim=1))
            self.book.set(lat.b.name, pm[f"a2c_network.pnn.actors.{k}.2.bias"])
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
