"""PULSE VAE policy network (``network: amp_z``) on the gfx950 kernels.

Mirrors AMPZBuilder.Network, phc/learning/amp_network_z_builder.py (z_type "vae", learned prior,
non-RNN branch) with the sizes of learning/im_z_fit.yaml + env/env_im_vae.yaml:

  encoder   z_mlp: obs(934) -> [1536,1024,512] SiLU -> Linear(160)            :469-492
            z_mu / z_logvar: Linear(160 -> 32) each                          :507-511
  prior     z_prior: self_obs(358) -> [1536,1024,512] SiLU; z_prior_mu / z_prior_logvar (512 -> 32)   :513-518, 226-241
  decoder   actor_mlp: cat(self_obs, z)(390) -> [3096,2048,1024] SiLU; mu (1024 -> 69)                :422-467
  critic    critic_z_mlp: obs -> [1536,1024,512] SiLU -> Linear(32); critic_mlp: cat(self_obs, .) -> ... -> value   :559-580, 326-339

The heavy part (every Linear, forward and backward) runs as launch plans of the fp32 MFMA GEMM
(learning/graph.py).  The head-level algebra of form_embedding (:82-121: log-var clamp [-5, vae_var_clamp_max],
re-parameterisation) and of the losses is tiny ((B,32)/(B,69) tensors) and is done with ordinary
tensor ops by the caller between the plans; its gradients are fed back through ``seed_*``.

Physical layout notes: z_mu|z_logvar and z_prior_mu|z_prior_logvar are each ONE stacked (64, .) matrix
(one GEMM, no gradient accumulation at their shared input); cat(self_obs, z) is a 392-float buffer with
self_obs at columns 0..357 and z at 360..391 (16-byte aligned segment), the first decoder / critic layer's
weight carries the same column layout.  ``state_dict`` speaks the reference's names and shapes.
"""
import torch

from .. import kernels as K
from .._lib import ACT_NONE
from .graph import Linear, MlpGraph, ParamBook, init_linear_, r4


class AMPZNetwork:
    def __init__(self, params, *, actions_num, self_obs_size, task_obs_size, task_obs_size_detail, device="cuda:0", split_k=8):
        self.device = torch.device(device)
        d = task_obs_size_detail
        self.embedding_size = int(d.get("embedding_size", 32))
        self.z_type = d.get("z_type", "vae")
        if self.z_type != "vae":
            raise NotImplementedError("only z_type 'vae' (the shipped PULSE configuration) is built")
        self.use_vae_prior = bool(d.get("use_vae_prior", True))
        if not self.use_vae_prior:
            raise NotImplementedError("fixed / absent priors are not built (use_vae_prior: True in env_im_vae.yaml)")
        self.use_vae_clamped_prior = bool(d.get("use_vae_clamped_prior", True))
        self.vae_var_clamp_max = float(d.get("vae_var_clamp_max", 2))
        self.self_obs_size, self.task_obs_size = int(self_obs_size), int(task_obs_size)
        self.obs_size = self.self_obs_size + self.task_obs_size
        self.in_pitch = (self.obs_size + 31) // 32 * 32
        self.actions_num = int(actions_num)
        self.a_pitch = r4(self.actions_num)
        self.units = [int(u) for u in params["mlp"]["units"]]
        self.task_units = [int(u) for u in params["task_mlp"]["units"]]
        self.act = K.ACTIVATIONS[params["mlp"]["activation"]]
        self.task_act = K.ACTIVATIONS[params["task_mlp"]["activation"]]
        if not params.get("separate", False):
            raise NotImplementedError("separate: True required")
        si = params["space"]["continuous"].get("sigma_init", {"val": 0.0})
        self.sigma = torch.full((self.actions_num,), float(si.get("val", 0.0)), dtype=torch.float32, device=self.device)
        self.split_k = split_k
        self.z_col = r4(self.self_obs_size)                     # 360: where z starts inside cat(self_obs, z)
        self.cat_width = self.z_col + self.embedding_size
        self.cat_map = list(range(self.self_obs_size)) + list(range(self.z_col, self.z_col + self.embedding_size))
        self._graphs = {}
        self.book = None
        g = self._build(1, register_only=True)                  # registers every parameter in construction order
        self.book.finalize()
        self.lins = g["lins"]
        self.reset_parameters()
        self.training = True

    # ------------------------------------------------------------------ construction
    def _build(self, m, register_only=False, x=None):
        first = self.book is None
        if first:
            self.book = ParamBook(self.device, self.split_k)
        book = self.book if first else _Replay(self.book)     # later graphs re-use the registered parameters
        E, S, T, U = self.embedding_size, self.self_obs_size, self.task_units, self.units
        g = MlpGraph(book, m)
        g.buffer("x", self.obs_size, tensor=x if x is not None else torch.zeros(m, self.in_pitch, device=self.device))
        g.buffer("ain", self.cat_width)
        g.buffer("cin", self.cat_width)
        lins = {}
        zc = (self.z_col, self.cat_width, ACT_NONE, None, 0)
        # A2CBuilder order: actor_mlp, critic_mlp, value, mu (network_builder.py:245-261)
        lins["actor_mlp"] = g.mlp(book, "a2c_network.actor_mlp", "ain", S + E, U, self.act, ["a1", "a2", "a3"][:len(U)], colmap=self.cat_map,
                                  first_grad_ranges=[zc], tag="dec")
        lins["critic_mlp"] = g.mlp(book, "a2c_network.critic_mlp", "cin", S + E, U, self.act, ["c1", "c2", "c3"][:len(U)], colmap=self.cat_map,
                                   first_grad_ranges=[zc], tag="critic")
        last_a, last_c = ["a1", "a2", "a3"][len(U) - 1], ["c1", "c2", "c3"][len(U) - 1]
        g.buffer("value", 1)
        g.buffer("mu", self.actions_num)
        lins["value"] = g.linear(Linear(book, "a2c_network.value", U[-1], 1), last_c, "value", grad_ranges=[(0, U[-1], self.act, last_c, 0)], tag="critic")
        lins["mu"] = g.linear(Linear(book, "a2c_network.mu", U[-1], self.actions_num), last_a, "mu", grad_ranges=[(0, U[-1], self.act, last_a, 0)], tag="dec")
        # _build_z_mlp (:469-521): encoder, its heads, the learned prior and its heads
        lins["z_mlp"] = g.mlp(book, "a2c_network.z_mlp", "x", self.obs_size, T, self.task_act, ["e1", "e2", "e3"][:len(T)],
                              final_linear=5 * E, final_dst="zenc", tag="enc")
        g.buffer("zheads", 2 * E)
        lins["zheads"] = g.linear(Linear(book, "a2c_network.z_heads", 5 * E, 2 * E), "zenc", "zheads", grad_ranges=[(0, 5 * E, ACT_NONE, None, 0)], tag="enc")
        lins["z_prior"] = g.mlp(book, "a2c_network.z_prior", "x", S, T, self.task_act, ["p1", "p2", "p3"][:len(T)], tag="prior")
        last_p = ["p1", "p2", "p3"][len(T) - 1]
        g.buffer("pheads", 2 * E)
        lins["pheads"] = g.linear(Linear(book, "a2c_network.z_prior_heads", T[-1], 2 * E), last_p, "pheads",
                                  grad_ranges=[(0, T[-1], self.task_act, last_p, 0)], tag="prior")
        # _build_critic_z_mlp (:559-580): its final Linear writes the z slot of the critic's concat buffer
        lins["critic_z_mlp"] = g.mlp(book, "a2c_network.critic_z_mlp", "x", self.obs_size, T, self.task_act, ["q1", "q2", "q3"][:len(T)],
                                     final_linear=E, final_dst="cin", final_dst_col=self.z_col, tag="critic_z")
        return {"graph": g, "lins": lins}

    def graph(self, m, x=None):
        """Buffers + plans for a batch of m rows; ``x`` (m, in_pitch) is the normalised observation buffer."""
        key = (m, x.data_ptr() if x is not None else 0)
        if key in self._graphs:
            return self._graphs[key]
        built = self._build(m, x=x)
        g = built["graph"]
        out = {"g": g, "x": g.act_bufs["x"],
               "fwd_enc": g.forward_plan({"enc"}), "fwd_prior": g.forward_plan({"prior"}), "fwd_dec": g.forward_plan({"dec"}),
               "fwd_critic": _concat_plans(g.forward_plan({"critic_z"}), g.forward_plan({"critic"})),
               "bwd_dec": g.backward_plan({"dec"}), "bwd_enc": g.backward_plan({"enc"}), "bwd_prior": g.backward_plan({"prior"}),
               "bwd_critic": _concat_plans(g.backward_plan({"critic"}), g.backward_plan({"critic_z"}))}
        self._graphs[key] = out
        return out

    # ------------------------------------------------------------------ parameters (reference names)
    def _names(self):
        """reference key -> (book name, row slice or None)"""
        E = self.embedding_size
        out = {}
        for p in self.book.params.values():
            n = p.name
            if n.startswith("a2c_network.z_heads"):
                kind = n.split(".")[-1]
                out[f"a2c_network.z_mu.{kind}"] = (n, slice(0, E))
                out[f"a2c_network.z_logvar.{kind}"] = (n, slice(E, 2 * E))
            elif n.startswith("a2c_network.z_prior_heads"):
                kind = n.split(".")[-1]
                out[f"a2c_network.z_prior_mu.{kind}"] = (n, slice(0, E))
                out[f"a2c_network.z_prior_logvar.{kind}"] = (n, slice(E, 2 * E))
            else:
                out[n] = (n, None)
        return out

    def _logical(self, name, sl, buf=None):
        p = self.book.params[name]
        v = self.book.get(name, buf)
        if name.endswith(".bias"):                          # bias row
            v = v.reshape(-1)
            return v[sl] if sl is not None else v
        return v[sl] if sl is not None else v

    def state_dict(self, buf=None):
        sd = {k: self._logical(n, sl, buf).clone() for k, (n, sl) in self._names().items()}
        sd["a2c_network.sigma"] = self.sigma.clone()
        return sd

    def gradients(self):
        return {k: self._logical(n, sl, self.book.grad).clone() for k, (n, sl) in self._names().items()}

    def load_state_dict(self, sd, strict=True):
        stacked = {}
        for k, (n, sl) in self._names().items():
            if k not in sd:
                if strict:
                    raise KeyError(k)
                continue
            v = sd[k].to(self.device, torch.float32)
            if sl is None:
                self.book.set(n, v)
            else:
                stacked.setdefault(n, {})[sl.start] = v
        for n, parts in stacked.items():
            p = self.book.params[n]
            is_bias = n.endswith(".bias")
            full = torch.cat([parts[k].reshape(1, -1) if is_bias else parts[k].reshape(-1, p.cols) for k in sorted(parts)],
                             dim=1 if is_bias else 0)
            self.book.set(n, full)
        if "a2c_network.sigma" in sd:
            self.sigma.copy_(sd["a2c_network.sigma"].to(self.device, torch.float32))

    def reset_parameters(self, generator=None):
        for group in self.lins.values():
            for lin in (group if isinstance(group, list) else [group]):
                init_linear_(self.book, lin, generator)

    def parameters_count(self):
        return self.book.n_flat

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ head algebra (form_embedding :82-121, compute_prior :226-241)
    def split_heads(self, heads):
        E = self.embedding_size
        mu, logvar = heads[:, :E], heads[:, E:2 * E]
        if self.use_vae_clamped_prior:
            logvar = torch.clamp(logvar, min=-5, max=self.vae_var_clamp_max)
        return mu, logvar


class _Replay:
    """ParamBook look-alike handed to graph builders AFTER the real book exists: ``add`` returns the
    already-registered parameter instead of allocating (so every batch size shares one flat buffer)."""

    def __init__(self, book):
        self._book = book

    def add(self, name, rows, cols, colmap=None, pitch_align=4):
        return self._book.params[name]

    def __getattr__(self, k):
        return getattr(self._book, k)


def _concat_plans(a, b):
    p = K.Plan()
    p.ops = a.ops + b.ops
    return p
