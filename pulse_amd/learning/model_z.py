"""Adapter that gives the PULSE VAE network (network_z.AMPZNetwork) the model interface CommonAgent /
AMPAgent drive: workspace(m, train) / forward / eval_critic / backward over one flat parameter buffer.

Forward of the policy follows AMPZBuilder.Network.eval_actor (amp_network_z_builder.py:422-467):
encoder plan -> form_embedding (log-var clamp, re-parameterisation with fresh N(0,1) noise, :82-121) ->
cat(self_obs, z) -> decoder plan -> mu.  The (B,32) head algebra runs as ordinary tensor ops on a detached
leaf so that the SAME formulas provide the gradients (d z / d mu, d z / d logvar with the clamp mask) when
the decoder's input gradient comes back from the GEMM plans.
"""
import torch

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

    def _embed(self, ws, need_grad):
        """form_embedding on the encoder heads; writes z and self_obs into the decoder's concat buffer."""
        net, g = self.net, ws["g"]
        E, S, zc = net.embedding_size, net.self_obs_size, net.z_col
        heads = g.act_bufs["zheads"].detach()
        if need_grad:
            heads = heads.clone().requires_grad_(True)
        with torch.enable_grad() if need_grad else torch.no_grad():
            vae_mu, vae_logvar = net.split_heads(heads)
            eps = ws["z_noise"] if ws["z_noise"] is not None else torch.randn(heads.shape[0], E, device=self.device, generator=self.generator)
            z = vae_mu + torch.exp(0.5 * vae_logvar) * eps
        ain = g.act_bufs["ain"]
        ain[:, :S].copy_(ws["x"][:, :S])
        ain[:, zc:zc + E].copy_(z.detach())
        ws.update({"heads_leaf": heads, "vae_mu": vae_mu, "vae_log_var": vae_logvar, "z": z, "eps": eps})

    def forward_actor(self, ws, m, need_grad=None):
        need_grad = self.training if need_grad is None else need_grad
        ws["G"]["fwd_enc"].run()
        self._embed(ws, need_grad)
        ws["G"]["fwd_dec"].run()

    def eval_critic(self, ws, m):
        g, S = ws["g"], self.net.self_obs_size
        g.act_bufs["cin"][:, :S].copy_(ws["x"][:, :S])
        ws["G"]["fwd_critic"].run()

    def forward(self, ws, m):
        self.forward_actor(ws, m)
        self.eval_critic(ws, m)

    def compute_prior(self, ws, need_grad=False):
        """compute_prior (:226-241) -> (prior_mu, prior_logvar) as tensors on a detached leaf."""
        ws["G"]["fwd_prior"].run()
        ph = ws["g"].act_bufs["pheads"].detach()
        if need_grad:
            ph = ph.clone().requires_grad_(True)
        with torch.enable_grad() if need_grad else torch.no_grad():
            pm, pv = self.net.split_heads(ph)
        ws["prior_leaf"] = ph
        return pm, pv

    # ---- backward
    def backward_actor(self, ws, extra_loss=None):
        """d loss/d mu is already in ws['dmu'].  Back-propagates decoder -> z -> encoder; ``extra_loss`` is an
        optional scalar built from ws['vae_mu'] / ws['vae_log_var'] (e.g. the KL term) whose gradient is added
        at the heads."""
        g, net = ws["g"], self.net
        E, zc = net.embedding_size, net.z_col
        ws["G"]["bwd_dec"].run()
        dz = g.grad("ain")[:, zc:zc + E]
        tensors, grads = [ws["z"]], [dz]
        if extra_loss is not None:
            tensors.append(extra_loss)
            grads.append(None)
        torch.autograd.backward(tensors, grads)
        g.grad("zheads").copy_(ws["heads_leaf"].grad)
        ws["G"]["bwd_enc"].run()

    def backward_prior(self, ws):
        ws["g"].grad("pheads").copy_(ws["prior_leaf"].grad)
        ws["G"]["bwd_prior"].run()

    def backward(self, ws, m, grad_scale=1.0):
        """PPO backward: actor (through the VAE) + critic; weight gradients of untouched sub-nets stay zero."""
        self.book.slabs.zero_()
        self.backward_actor(ws)
        ws["G"]["bwd_critic"].run()
        return self.book.reduce_grads(grad_scale)
