"""The terrain-task policy network (``network: amp_sept``) on the gfx950 kernels.

Mirrors AMPSeptBuilder.Network, phc/learning/amp_network_sept_builder.py:19-165 (registered at run_hydra.py:260; config
phc/data/cfg/learning/pulse_z_terrain.yaml:12-44): the observation is [self obs | task obs], the task part (trajectory samples + height
map, 1044 floats) goes through a task MLP ([512, 256] SiLU, both layers activated) and the actor / critic MLPs ([2048, 1024, 512] SiLU)
read cat(self_obs, task_out) (358 + 256).

One reference quirk is structural and mirrored: with ``separate: True`` the constructor calls ``_build_task_mlp()`` twice
(:33-36) and both calls assign ``self._task_mlp`` -- so there is ONE task MLP, shared by eval_actor (:92-101) and eval_critic
(:77-90), and its weights receive the sum of both gradients.  (The first instance is garbage-collected; only its RNG draws remain.)
The 'people' point-net branch (:45-60, 152-165) belongs to the crowd variant (``_divide_group``), which no shipped config enables.

Physical layout (learning/graph.py): the task MLP reads the observation buffer from column 356 (16-byte aligned; the two self-observation
columns in front carry zero weights through the layer's column map), its last layer writes straight into columns 360..615 of the actor's
concat buffer, a copy places the same 256 values (and the self observation) in the critic's; backward runs actor and critic down to
their concat buffers (SiLU derivative of the task output fused into both input-gradient GEMMs), adds the two task-output gradients and
continues through the task MLP.
"""
import torch

from .. import kernels as K
from .graph import Linear, MlpGraph, ParamBook, init_linear_, r4
from .network_z import _Replay


class AMPSeptNetwork:
    def __init__(self, params, *, actions_num, self_obs_size, task_obs_size, task_obs_size_detail, device="cuda:0", split_k=8):
        self.device = torch.device(device)
        d = dict(task_obs_size_detail or {})
        if "people" in d:
            raise NotImplementedError("amp_sept's point-net branch ('people': the crowd variant of the terrain task) is not built")
        if "traj" not in d or "heightmap" not in d:
            raise ValueError("amp_sept needs task_obs_size_detail with 'traj' and 'heightmap' (amp_network_sept_builder.py:107)")
        if d["traj"] + d["heightmap"] != task_obs_size:
            raise ValueError("task_obs_size_detail does not add up to task_obs_size")
        if not params.get("separate", False):
            raise NotImplementedError("separate: True required")
        self.self_obs_size, self.task_obs_size = int(self_obs_size), int(task_obs_size)
        self.obs_size = self.self_obs_size + self.task_obs_size
        self.in_pitch = (self.obs_size + 31) // 32 * 32
        self.actions_num = int(actions_num)
        self.a_pitch = r4(self.actions_num)
        self.units = [int(u) for u in params["mlp"]["units"]]
        self.task_units = [int(u) for u in params["task_mlp"]["units"]]
        self.act = K.ACTIVATIONS[params["mlp"]["activation"]]
        self.task_act = K.ACTIVATIONS[params["task_mlp"]["activation"]]
        si = params["space"]["continuous"].get("sigma_init", {"val": 0.0})
        self.sigma = torch.full((self.actions_num,), float(si.get("val", 0.0)), dtype=torch.float32, device=self.device)
        self.split_k = split_k
        S, TU = self.self_obs_size, self.task_units[-1]
        self.t_col = r4(S)                                       # 360: where the task embedding starts inside cat(self_obs, task_out)
        self.cat_width = self.t_col + TU
        self.cat_map = list(range(S)) + list(range(self.t_col, self.t_col + TU))
        self.task_src_col = S // 4 * 4                           # 356: aligned start of the task MLP's input window
        self.task_map = list(range(S - self.task_src_col, S - self.task_src_col + self.task_obs_size))
        self._graphs = {}
        self.book = None
        g = self._build(1)
        self.book.finalize()
        self.lins = g["lins"]
        self.reset_parameters()
        self.training = True

    def _build(self, m, x=None):
        first = self.book is None
        if first:
            self.book = ParamBook(self.device, self.split_k)
        book = self.book if first else _Replay(self.book)
        S, U, T = self.self_obs_size, self.units, self.task_units
        g = MlpGraph(book, m)
        g.buffer("x", self.obs_size, tensor=x if x is not None else torch.zeros(m, self.in_pitch, device=self.device))
        g.buffer("ain", self.cat_width)
        g.buffer("cin", self.cat_width)
        lins = {}
        names_a, names_c = ["a1", "a2", "a3", "a4"][:len(U)], ["c1", "c2", "c3", "c4"][:len(U)]
        # the task output sits in columns t_col.. of BOTH concat buffers; its pre-activation is kept beside the actor's copy
        tc = (self.t_col, self.cat_width, self.task_act, "ain", 0)
        # A2CBuilder order: actor_mlp, critic_mlp, value, mu (network_builder.py:245-261), then the task MLP (amp_network_sept_builder.py:33-36)
        lins["actor_mlp"] = g.mlp(book, "a2c_network.actor_mlp", "ain", S + T[-1], U, self.act, names_a, colmap=self.cat_map, first_grad_ranges=[tc], tag="actor")
        lins["critic_mlp"] = g.mlp(book, "a2c_network.critic_mlp", "cin", S + T[-1], U, self.act, names_c, colmap=self.cat_map, first_grad_ranges=[tc],
                                   tag="critic")
        g.buffer("value", 1)
        g.buffer("mu", self.actions_num)
        lins["value"] = g.linear(Linear(book, "a2c_network.value", U[-1], 1), names_c[-1], "value", grad_ranges=[(0, U[-1], self.act, names_c[-1], 0)], tag="critic")
        lins["mu"] = g.linear(Linear(book, "a2c_network.mu", U[-1], self.actions_num), names_a[-1], "mu", grad_ranges=[(0, U[-1], self.act, names_a[-1], 0)], tag="actor")
        # _task_mlp = Sequential(Linear, SiLU, Linear, SiLU): every layer activated; the last one writes the actor's concat slot
        task = []
        cur, cur_w = "x", self.task_obs_size
        for i, u in enumerate(T):
            last = i == len(T) - 1
            lin = Linear(book, f"a2c_network._task_mlp.{2 * i}", cur_w, u, self.task_act, self.task_map if i == 0 else None)
            dst = "ain" if last else f"t{i + 1}"
            if not last:
                g.buffer(dst, u)
            gr = None if i == 0 else [(0, cur_w, self.task_act, cur, 0)]
            g.linear(lin, cur, dst, src_col=self.task_src_col if i == 0 else 0, dst_col=self.t_col if last else 0, grad_ranges=gr, tag="task")
            task.append(lin)
            cur, cur_w = dst, u
        lins["task_mlp"] = task
        return {"graph": g, "lins": lins}

    def graph(self, m, x=None):
        key = (m, x.data_ptr() if x is not None else 0)
        if key in self._graphs:
            return self._graphs[key]
        g = self._build(m, x=x)["graph"]
        out = {"g": g, "x": g.act_bufs["x"], "fwd_task": g.forward_plan({"task"}), "fwd_actor": g.forward_plan({"actor"}),
               "fwd_critic": g.forward_plan({"critic"}), "bwd_actor": g.backward_plan({"actor"}), "bwd_critic": g.backward_plan({"critic"}),
               "bwd_task": g.backward_plan({"task"})}
        self._graphs[key] = out
        return out

    # ------------------------------------------------------------------ parameters (reference names)
    def state_dict(self, buf=None):
        sd = {}
        for p in self.book.params.values():
            v = self.book.get(p.name, buf)
            sd[p.name] = (v.reshape(-1) if p.name.endswith(".bias") else v).clone()
        sd["a2c_network.sigma"] = self.sigma.clone()
        return sd

    def gradients(self):
        sd = self.state_dict(self.book.grad)
        sd.pop("a2c_network.sigma")
        return sd

    def load_state_dict(self, sd, strict=True):
        for p in self.book.params.values():
            if p.name not in sd:
                if strict:
                    raise KeyError(p.name)
                continue
            self.book.set(p.name, sd[p.name].to(self.device, torch.float32))
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


class AMPSeptModel:
    """The model interface CommonAgent / AMPAgent drive (workspace / forward / eval_critic / backward over one flat buffer)."""

    def __init__(self, params, *, actions_num, self_obs_size, task_obs_size, task_obs_size_detail, device, split_k=8):
        self.net = AMPSeptNetwork(params, actions_num=actions_num, self_obs_size=self_obs_size, task_obs_size=task_obs_size,
                                  task_obs_size_detail=task_obs_size_detail, device=device, split_k=split_k)
        n = self.net
        self.device, self.book = n.device, n.book
        self.flat, self.grad, self.n_flat = n.book.flat, n.book.grad, n.book.n_flat
        self.sigma, self.a_pitch, self.in_pitch, self.actions_num = n.sigma, n.a_pitch, n.in_pitch, n.actions_num
        self.training = True
        self.mixed_precision = False
        self._ws = {}

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

    def workspace(self, m, train):
        ws = self._ws.get(m)
        if ws is None:
            G = self.net.graph(m)
            g, A = G["g"], self.actions_num
            ws = {"G": G, "g": g, "x": G["x"], "mu": g.act_bufs["mu"][:, :A], "val": g.act_bufs["value"][:, :1],
                  "dmu": g.grad("mu")[:, :A], "dval": g.grad("value")[:, :1]}
            self._ws[m] = ws
        return ws

    def _task(self, ws, critic_only=False):
        """eval_task (:43-60) once, then cat([self_obs, task_out]) for both consumers (:84-86, 96-98)."""
        net, g = self.net, ws["g"]
        S, tc, cw = net.self_obs_size, net.t_col, net.cat_width
        ws["G"]["fwd_task"].run()
        ain, cin = g.act_bufs["ain"], g.act_bufs["cin"]
        ain[:, :S].copy_(ws["x"][:, :S])
        cin[:, :S].copy_(ws["x"][:, :S])
        cin[:, tc:cw].copy_(ain[:, tc:cw])

    def forward(self, ws, m):
        self._task(ws)
        ws["G"]["fwd_actor"].run()
        ws["G"]["fwd_critic"].run()

    def eval_critic(self, ws, m):
        self._task(ws)
        ws["G"]["fwd_critic"].run()

    supports_fused_sqnorm = True

    def backward(self, ws, m, grad_scale=1.0, sq_partials=None, on_bucket=None):
        """d loss / d (mu, value) are in ws['dmu'] / ws['dval'].  The shared task MLP receives the sum of the actor's and the critic's
        gradient at its output (both already carry the SiLU derivative from the fused input-gradient epilogues).  Every layer is visited,
        so the region reduce reads exactly the slabs the weight-gradient launches wrote (no zero fill of the slab buffer) and leaves the norm
        clip's sums of squares in ``sq_partials``."""
        net, g = self.net, ws["g"]
        tc, cw = net.t_col, net.cat_width
        ws["G"]["bwd_actor"].run()
        ws["G"]["bwd_critic"].run()
        g.grad("ain")[:, tc:cw].add_(g.grad("cin")[:, tc:cw])
        ws["G"]["bwd_task"].run()
        if "untouched" not in ws:
            ws["untouched"] = g.untouched_ranges({"actor", "critic", "task"})
        if not self.book.reduce_grads(grad_scale, untouched=ws["untouched"], sq_partials=sq_partials) and sq_partials is not None:
            K.sqnorm_partial(self.book.grad, self.book.n_flat, sq_partials)
        return self.book.grad
