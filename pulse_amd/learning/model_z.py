"""Adapter that gives the PULSE VAE network (network_z.AMPZNetwork) the model interface CommonAgent /
AMPAgent drive: workspace(m, train) / forward / eval_critic / backward over one flat parameter buffer.

Forward of the policy follows AMPZBuilder.Network.eval_actor (amp_network_z_builder.py:422-467):
encoder plan -> form_embedding (log-var clamp, re-parameterisation with fresh N(0,1) noise, :82-121) ->
cat(self_obs, z) -> decoder plan -> mu.  The (B,32) head algebra is two fused launches (pulse_amd/csrc/vae_head.hip):
pulse_vae_embed on the way forward, pulse_vae_head_backward (d z / d mu, d z / d logvar with the clamp mask, plus the KL / AR(1) /
regulariser terms of _optimize_kin) when the decoder's input gradient comes back from the GEMM plans.  No autograd tape.
"""
import torch

from .. import kernels as K

from .network_z import AMPZNetwork


class AMPZModel:
    def __init__(self, params, *, actions_num, self_obs_size, task_obs_size, task_obs_size_detail, device, split_k=8, generator=None):
        self.net = AMPZNetwork(params, actions_num=actions_num, self_obs_size=self_obs_size, task_obs_size=task_obs_size,
                               task_obs_size_detail=task_obs_size_detail, device=device, split_k=split_k)
        n = self.net
        self.device = n.device
        self.book = n.book
        self.flat, self.grad, self.n_flat = n.book.flat, n.book.grad, n.book.n_flat
        self.sigma, self.a_pitch, self.in_pitch, self.actions_num = n.sigma, n.a_pitch, n.in_pitch, n.actions_num
        self.embedding_size = n.embedding_size
        self.generator = generator
        self.training = True
        self._ws = {}

    # ---- nn.Module-like surface
    def parameters_count(self):
        return self.n_flat

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def is_rnn(self):
        return False

    def state_dict(self):
        return self.net.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.net.load_state_dict(sd, strict)

    # ---- workspaces
    def workspace(self, m, train):
        ws = self._ws.get(m)
        if ws is None:
            G = self.net.graph(m)
            g, A = G["g"], self.actions_num
            ws = {"G": G, "g": g, "x": G["x"], "mu": g.act_bufs["mu"][:, :A], "val": g.act_bufs["value"][:, :1],
                  "dmu": g.grad("mu")[:, :A], "dval": g.grad("value")[:, :1], "z_noise": None}
            self._ws[m] = ws
        return ws

    def _embed(self, ws):
        """form_embedding (amp_network_z_builder.py:79-121) on the encoder heads in one launch: clamp, re-parameterise, and place z and the
        self observation in the decoder's (and the critic's) concat buffer.  The noise is ws['z_noise'] when a test injected it, a fresh
        draw otherwise; z = mu in test mode."""
        net, g = self.net, ws["g"]
        E, S = net.embedding_size, net.self_obs_size
        heads = g.act_bufs["zheads"]
        m = heads.shape[0]
        if getattr(self, "deterministic_z", False):
            eps = None
        elif ws["z_noise"] is not None:
            eps = ws["z_noise"]
        else:
            eps = ws.get("_eps_buf")
            if eps is None:
                eps = ws["_eps_buf"] = torch.empty(m, E, device=self.device)
            eps.normal_(generator=self.generator)
        K.vae_embed(heads, ws["x"], g.act_bufs["ain"], rows=m, embedding_size=E, self_obs_size=S, z_col=net.z_col, eps=eps, cin=g.act_bufs["cin"],
                    clamp=net.use_vae_clamped_prior, clamp_max=net.vae_var_clamp_max)
        ws["eps"] = eps

    def forward_actor(self, ws, m, need_grad=None):
        ws["G"]["fwd_enc"].run()
        self._embed(ws)
        ws["G"]["fwd_dec"].run()

    def eval_critic(self, ws, m):
        g, S = ws["g"], self.net.self_obs_size
        g.act_bufs["cin"][:, :S].copy_(ws["x"][:, :S])
        ws["G"]["fwd_critic"].run()

    def forward(self, ws, m):
        self.forward_actor(ws, m)                                    # (_embed also fills the critic's self-observation columns)
        ws["G"]["fwd_critic"].run()

    def compute_prior(self, ws, need_grad=False):
        """compute_prior (:226-241): runs the prior MLP; the raw heads [mu | logvar] stay in the graph buffer 'pheads' (split_heads clamps)."""
        ws["G"]["fwd_prior"].run()
        return self.net.split_heads(ws["g"].act_bufs["pheads"])

    # ---- backward
    def backward_actor(self, ws, kin=None):
        """d loss/d mu is already in ws['dmu'].  Back-propagates decoder -> z -> encoder.  ``kin``: the head-level terms of _optimize_kin
        (dict c_kl, c_ar1, c_regu, progress, horizon) whose gradients join the re-parameterisation path at the encoder / prior heads; one
        launch of pulse_vae_head_backward either way."""
        g, net = ws["g"], self.net
        E, zc = net.embedding_size, net.z_col
        ws["G"]["bwd_dec"].run()
        dz = g.grad("ain")[:, zc:zc + E]
        heads = g.act_bufs["zheads"]
        kw = dict(rows=heads.shape[0], embedding_size=E, eps=ws["eps"], dz=dz, clamp=net.use_vae_clamped_prior, clamp_max=net.vae_var_clamp_max)
        if kin is not None:
            kw.update(pheads=g.act_bufs["pheads"], dpheads=g.grad("pheads"), progress=kin.get("progress"), horizon=kin.get("horizon", 1),
                      c_kl=kin["c_kl"], c_ar1=kin.get("c_ar1", 0.0), c_regu=kin.get("c_regu", 0.0))
        K.vae_head_backward(heads, g.grad("zheads"), **kw)
        ws["G"]["bwd_enc"].run()

    def backward_prior(self, ws):
        ws["G"]["bwd_prior"].run()                                  # d loss / d prior heads was written by backward_actor(kin=...)

    supports_fused_sqnorm = True

    def backward(self, ws, m, grad_scale=1.0, sq_partials=None, on_bucket=None):
        """PPO backward: actor (through the VAE) + critic; weight gradients of untouched sub-nets (the prior: kin pass only) are zero.
        ``sq_partials``: the gradient reduce also leaves the norm clip's per-block sums of squares there; returns True when it did."""
        self.backward_actor(ws)
        ws["G"]["bwd_critic"].run()
        done = self.book.reduce_grads(grad_scale, untouched=self._untouched(ws, ("dec", "enc", "critic", "critic_z")), sq_partials=sq_partials)
        if sq_partials is not None and not done:                # (the uniform reduce: more regions than one launch takes)
            K.sqnorm_partial(self.book.grad, self.book.n_flat, sq_partials)
        return self.book.grad

    def _untouched(self, ws, tags):
        key = ("untouched",) + tuple(tags)
        if key not in ws:
            ws[key] = ws["g"].untouched_ranges(set(tags))
        return ws[key]
