"""Actor / critic MLP pair on the gfx950 GEMM kernel, with an explicit (non-autograd) backward.

Mirrors the *MLP branch* of the reference network stack:
  A2CBuilder.Network / AMPBuilder.Network   phc/learning/network_builder.py:188-291,
                                            phc/learning/amp_network_builder.py:19-40,58-216
  ModelA2CContinuousLogStd (rl_games 3P)    SURVEY.md Appendix B
i.e. ``separate: True`` actor and critic MLPs of identical shape, a linear ``mu`` head, a
state-independent non-learned ``sigma`` (``fixed_sigma: True, learn_sigma: False``,
learning/im.yaml:21-25) and a linear ``value`` head.  Checkpoint key names are the reference's
(``a2c_network.actor_mlp.0.weight`` ...) because ``network_loader.py:76-176`` treats them as a
wire format.

MI355X-first layout (what differs from a stack of nn.Linear):
  * ALL parameters live in one flat fp32 buffer (one fused clip+Adam launch, one RCCL all-reduce);
    same-shaped actor / critic layers are adjacent so one batched GEMM launch serves both nets;
  * layer 1 of both nets is ONE GEMM of N = 2*u1 over the shared normalised observation;
  * the observation pitch is padded to a multiple of 32 floats (934 -> 960), weights carry the
    matching zero columns;
  * backward is hand-derived: dX GEMMs fuse the activation derivative, dW GEMMs split the batch
    (reduction) dimension into S slabs that are summed by one deterministic reduce launch which
    also yields the flat gradient ready for all-reduce / clipping.
"""
import math

import os

import torch

from .. import kernels as K
from .._lib import ACT_RELU, ACT_SILU, EPI_RELU_GRAD, EPI_SILU_GRAD, GEMM_OUT_CONTIG


# PULSE_RELU_BITMASK=0: the bf16-storage path's relu-grad launches re-read the bf16 activations instead of the forward's sign bytes (A/B switch; same bits)
RELU_BITMASK = os.environ.get("PULSE_RELU_BITMASK", "1") != "0"
# PULSE_RELU_BITMASK_F32=0: the fp32 path's input-gradient launches re-read the activations instead of the forward's sign BITS (pulse_gemm_desc.relu_mask;
# A/B switch, same bits).  History: the 128 x 128 / 64 x 128 kernel's store loop that consumed the bits returned garbage beside another
# stream's / process's GEMM waves; that loop is gone (the bits are expanded into the aux path's registers), the probes and the stress runs are clean, and the
# masks are on again -- except for a policy network with a concurrent chain beside it (A2CNetwork.concurrent_chain), which keeps the re-read ("2" forces them
# there too: stress tests).
RELU_BITMASK_F32 = os.environ.get("PULSE_RELU_BITMASK_F32", "1") in ("1", "2")
RELU_BITMASK_F32_FORCE = os.environ.get("PULSE_RELU_BITMASK_F32", "0") == "2"      # (stress tests: also beside a concurrent chain)


def _r4(x):
    return (x + 3) // 4 * 4


def _r32(x):
    return (x + 31) // 32 * 32


class A2CNetwork:
    """Separate actor/critic MLP with mu / value heads.  Parameters: flat buffer ``self.flat``."""

    def __init__(self, params, *, actions_num, input_shape, value_size=1, device="cuda:0", split_k=8):
        self.device = torch.device(device)
        self._load(params)
        if not self.separate:
            raise NotImplementedError("only `separate: True` actor / critic networks are supported")
        if value_size != 1:
            raise NotImplementedError("value_size != 1")
        self.actions_num = int(actions_num)
        self.value_size = 1
        self.in_dim = int(input_shape[0] if isinstance(input_shape, (tuple, list)) else input_shape)
        self.in_pitch = _r32(self.in_dim)
        self.units = [int(u) for u in self.units]
        if any(u % 4 for u in self.units) or not self.units:
            raise NotImplementedError("MLP unit sizes must be non-empty multiples of 4")
        self.act = K.ACTIVATIONS[self.activation]
        if self.act not in (ACT_RELU, ACT_SILU):
            raise NotImplementedError(f"activation {self.activation!r} (relu / silu supported)")
        self.split_k = int(split_k)
        self.a_pitch = _r4(self.actions_num)
        self._build_layout()
        self.flat = torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
        self.sigma = torch.zeros(self.actions_num, dtype=torch.float32, device=self.device)   # log-std, non-learned
        self._slabs = None
        self.n_slabs = 2 * self.split_k      # allocated gradient slabs (the upper layers' fp32 weight gradients may use all of them, see workspace())
        self._slab_mode = None               # storage mode (fp32 / bf16) of the plan that last wrote the slabs
        self._flat16 = self._wt16 = None     # bf16 image of the flat parameters / transposed upper-layer weights (mixed_precision, built on demand)
        self._ws = {}
        # PULSE_L1_PLANAR=1: layer-1 forward on the planar GEMM (gemm_x3p.hip; same six-product arithmetic), its input planes written by the
        # normaliser.  OFF by default -- measured in situ on cfg2 (A/B in one gpurun call): the GEMM itself gains 11 % (196.6 vs
        # 176.1 TFLOP/s on the layer-1 forward), the 6 bytes of planes per element the normaliser writes beside its fp32 output and the weight
        # re-split cost more: 79.7 vs 78.8 ms per epoch.
        self.l1_planar = os.environ.get("PULSE_L1_PLANAR", "0") == "1" and K.F32_MODE == "x3"
        self._w1p = None
        self.training = True
        self.concurrent_chain = False     # True: another stream's GEMMs run beside this network's training pass (AMPAgent's discriminator chain): no fp32 bit masks (workspace)
        self.mixed_precision = False      # True: the TRAINING forward / backward run on the bf16 MFMA (amp_agent.py:671 autocast); inference stays fp32
        self.reset_parameters()

    # ------------------------------------------------------------------ config (network_builder.py:461-502)
    def _load(self, params):
        self.separate = params.get("separate", False)
        self.units = params["mlp"]["units"]
        self.activation = params["mlp"]["activation"]
        self.initializer = params["mlp"].get("initializer", {"name": "default"})
        if params["mlp"].get("d2rl", False) or params.get("normalization", None) is not None:
            raise NotImplementedError("d2rl / normalization layers are not used by the PULSE configs")
        if "rnn" in params or "cnn" in params:
            raise NotImplementedError("rnn / cnn branches are out of scope (no shipped config enables them)")
        space = params.get("space", {}).get("continuous")
        if space is None:
            raise NotImplementedError("continuous action space required")
        self.space_config = space
        if not space.get("fixed_sigma", True) or space.get("learn_sigma", True):
            raise NotImplementedError("only fixed_sigma: True / learn_sigma: False (state-independent constant sigma)")
        if space.get("mu_activation", "None") != "None" or space.get("sigma_activation", "None") != "None":
            raise NotImplementedError("mu / sigma activations other than None")
        if self.initializer.get("name", "default") != "default":
            raise NotImplementedError("only the `default` (nn.Linear) initializer")

    # ------------------------------------------------------------------ flat layout
    def _build_layout(self):
        u, L = self.units, len(self.units)
        off = 0
        self.w_off, self.b_off, self.in_w = [], [], []
        for l in range(L):
            k = self.in_pitch if l == 0 else u[l - 1]
            self.in_w.append(k)
            self.w_off.append(off)
            off += 2 * u[l] * k             # [actor rows | critic rows], pitch k
            self.b_off.append(off)
            off += 2 * u[l]
        # heads: two blocks of identical shape [A][uL] so ONE batched GEMM serves both:
        #   block 0 = mu.weight (A rows), block 1 row 0 = value.weight, rows 1.. stay zero (zero gradient, Adam keeps 0)
        self.head_rows = self.actions_num
        self.wh_off = off; off += 2 * self.head_rows * u[-1]
        off = _r4(off)
        self.bh_off = off; off += 2 * self.a_pitch
        self.wmu_off, self.wv_off = self.wh_off, self.wh_off + self.head_rows * u[-1]
        self.bmu_off, self.bv_off = self.bh_off, self.bh_off + self.a_pitch
        self.n_flat = _r4(off)

    def _w_view(self, buf, l, net):
        u, k = self.units[l], self.in_w[l]
        o = self.w_off[l] + net * u * k
        w = buf[o:o + u * k].view(u, k)
        return w[:, :self.in_dim] if l == 0 else w

    def _b_view(self, buf, l, net):
        u = self.units[l]
        o = self.b_off[l] + net * u
        return buf[o:o + u]

    def named_parameters(self, buf=None):
        """Reference key names -> views into the flat buffer (Sequential indices: Linear at 2*l)."""
        buf = self.flat if buf is None else buf
        out = {}
        for net, name in ((0, "actor_mlp"), (1, "critic_mlp")):
            for l in range(len(self.units)):
                out[f"a2c_network.{name}.{2 * l}.weight"] = self._w_view(buf, l, net)
                out[f"a2c_network.{name}.{2 * l}.bias"] = self._b_view(buf, l, net)
        uL = self.units[-1]
        out["a2c_network.value.weight"] = buf[self.wv_off:self.wv_off + uL].view(1, uL)          # row 0 of head block 1
        out["a2c_network.value.bias"] = buf[self.bv_off:self.bv_off + 1]
        out["a2c_network.mu.weight"] = buf[self.wmu_off:self.wmu_off + self.actions_num * uL].view(self.actions_num, uL)
        out["a2c_network.mu.bias"] = buf[self.bmu_off:self.bmu_off + self.actions_num]
        return out

    def named_gradients(self):
        return self.named_parameters(self.grad)

    def state_dict(self):
        sd = {k: v.clone() for k, v in self.named_parameters().items()}
        sd["a2c_network.sigma"] = self.sigma.clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        mine = self.named_parameters()
        for k, v in mine.items():
            if k not in sd:
                if strict:
                    raise KeyError(k)
                continue
            if tuple(sd[k].shape) != tuple(v.shape):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(v.shape)}")
            v.copy_(sd[k].to(self.device, torch.float32))
        if "a2c_network.sigma" in sd:
            self.sigma.copy_(sd["a2c_network.sigma"].to(self.device, torch.float32))

    def reset_parameters(self, generator=None):
        """nn.Linear default init (kaiming_uniform(a=sqrt 5) == U(+-1/sqrt(fan_in))), biases zeroed
        (network_builder.py:273-277), sigma = const_initializer(val) (im.yaml:21-23), drawn on the
        CPU in the reference's module-construction order so a shared seed reproduces the oracle."""
        def lin(o, i):
            w = torch.empty(o, i)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=generator)
            bound = 1 / math.sqrt(i)
            torch.empty(o).uniform_(-bound, bound, generator=generator)        # nn.Linear draws its bias too
            return w
        p = self.named_parameters()
        self.flat.zero_()
        for name in ("actor_mlp", "critic_mlp"):
            for l, uu in enumerate(self.units):
                i = self.in_dim if l == 0 else self.units[l - 1]
                p[f"a2c_network.{name}.{2 * l}.weight"].copy_(lin(uu, i))
        p["a2c_network.value.weight"].copy_(lin(1, self.units[-1]))
        p["a2c_network.mu.weight"].copy_(lin(self.actions_num, self.units[-1]))
        si = self.space_config.get("sigma_init", {"name": "const_initializer", "val": 0.0})
        self.sigma.fill_(float(si.get("val", 0.0)))

    def parameters_count(self):
        return self.n_flat

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def is_rnn(self):
        return False

    # ------------------------------------------------------------------ workspaces + launch plans
    def workspace(self, m, train):
        """Buffers and pre-built launch plans for a batch of m rows.
        ws['x']      (m, in_pitch)   normalised network input (written by the caller)
        ws['h'][l]   (m, 2*u_l)      hidden activations [actor | critic]
        ws['heads']  (m, 2*a_pitch)  [mu (A cols) .. | value at col a_pitch ..]
        train: ws['dheads'] (m, 2*a_pitch) = [d loss/d mu | d loss/d value at col a_pitch], ws['dh'][l]."""
        key = (m, bool(train), bool(self.mixed_precision))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, u = self.device, self.units
        e = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)     # one-time allocations: defined contents from the start
        ws = {"x": e(m, self.in_pitch), "h": [e(m, 2 * uu) for uu in u], "heads": torch.zeros(m, 2 * self.a_pitch, device=dev)}
        ws["mu"] = ws["heads"][:, :self.actions_num]                        # (m, A) view, row pitch 2*a_pitch
        ws["val"] = ws["heads"][:, self.a_pitch:self.a_pitch + 1]           # (m, 1) view
        if self.act == ACT_SILU:
            ws["z"] = [e(m, 2 * uu) for uu in u]
        ws["plan_fwd"] = self._plan_forward(ws, m, 0, 2)
        ws["plan_critic"] = self._plan_forward(ws, m, 1, 1)
        if self.l1_planar:
            # layer 1 on the planar GEMM (gemm_x3p.hip): the normaliser writes the three bf16 planes of its output next to the fp32 copy
            # (ws['xp']; the caller raises ws['xp_fresh'] when it did), the weight planes are re-split in front of every pass
            ws["xp"] = K.alloc_planes(m, self.in_dim, dev)
            ws["xp_fresh"] = False
            if self._w1p is None:
                self._w1p = K.alloc_planes(2 * u[0], self.in_dim, dev)
            ws["plan_fwd_planar"] = self._plan_forward(ws, m, 0, 2, planar=True)
            ws["plan_critic_planar"] = self._plan_forward(ws, m, 1, 1, planar=True)
        ws["b16"] = bool(train and self.mixed_precision and self.b16_storage_ok())
        if train:
            ws["dheads"] = torch.zeros(m, 2 * self.a_pitch, dtype=torch.float32, device=dev)
            ws["dmu"] = ws["dheads"][:, :self.actions_num]
            ws["dval"] = ws["dheads"][:, self.a_pitch:self.a_pitch + 1]
            if ws["b16"]:
                # bf16-STORAGE training passes (mixed_precision): the normalised input, every hidden activation and every gradient of this
                # workspace is a bf16 matrix in HBM (int16 bit patterns), the GEMMs are pulse_gemm_x3p(planes = 1).  Same arithmetic as the
                # fp32-storage bf16 kernel -- there every operand is rounded to bf16 on its way into LDS and every output leaves
                # bf16-representable -- at half the operand traffic, which is what bounded that kernel.
                i16 = lambda r, c: torch.zeros(r, c, dtype=torch.int16, device=dev)
                hp = _r32(self.head_rows)
                ws["x16"] = i16(m, self.in_pitch)
                ws["h16"] = [i16(m, 2 * uu) for uu in u]
                ws["dz16"] = [i16(m, 2 * uu) for uu in u]
                ws["dheads16"] = i16(m, 2 * hp)                              # [d mu (A) .. zero pad to hp | d value, zero pad]: written by ppo_loss
                ws["head_pitch16"] = hp
                if self._flat16 is None:
                    self._flat16 = i16(1, (self.n_flat + 7) // 8 * 8).view(-1)
                    self._wt16 = [None] + [i16(2 * u[l - 1], u[l]) for l in range(1, len(u))]     # W_l^T of both nets: (2, u_{l-1}, u_l)
                if self.act == ACT_RELU and RELU_BITMASK:
                    # [r6] one byte per row and 8 columns: written by the training forward, read by the input-gradient epilogues instead of h16
                    ws["hmask8"] = [K.alloc_relu_mask8(m, 2 * uu, dev) for uu in u]
                ws["plan_fwd_train"] = self._plan_forward_b16(ws, m)
            else:
                # ReLU nets: the training forward also records each hidden activation's sign bits (1 bit per element), and the input-gradient
                # launches mask with those instead of re-reading the fp32 activation matrix (134 MB per layer-1-wide launch at cfg2)
                if self.act == ACT_RELU and not self.mixed_precision and RELU_BITMASK_F32 and (RELU_BITMASK_F32_FORCE or not self.concurrent_chain):
                    ws["hmask"] = [K.alloc_relu_mask(m, 2 * uu, dev) for uu in u]
                    ws["plan_fwd_train"] = self._plan_forward(ws, m, 0, 2, masks=True)
                else:
                    ws["plan_fwd_train"] = self._plan_forward(ws, m, 0, 2, bf16=True) if self.mixed_precision else ws["plan_fwd"]
                ws["dh"] = [e(m, 2 * uu) for uu in u]
            if self._slabs is None:
                # twice split_k slabs: the upper layers' weight gradients are small outputs over a long reduction and fill the chip on the 256 x 256
                # x3 tile only with 16 slabs (profiles/r05_gemm_x3_wide_ab.txt: 2 x (512 x 1024) over 16384 rows 159 -> 136 us); a slab no launch
                # writes stays zero, and every reduce names the count it reads
                self._slabs = torch.zeros(self.n_slabs, self.n_flat, dtype=torch.float32, device=dev)
                self._bias_chunks = 64
                self._bias_scratch = torch.zeros(self._bias_chunks, 2 * max(max(u), self.a_pitch) + 8, dtype=torch.float32, device=dev)
                self._head_split = 32
                self._head_scratch = torch.zeros(self._head_split, self.bh_off - self.wh_off + 2 * self.a_pitch, dtype=torch.float32, device=dev)
            ws["plan_bwd"] = self._plan_backward_b16(ws, m) if ws["b16"] else self._plan_backward(ws, m)
        self._ws[key] = ws
        return ws

    def b16_storage_ok(self):
        """The bf16-storage GEMM reads reduction-contiguous operands in whole 32-deep k-tiles and 16-byte pieces: every hidden width has to
        be a multiple of 32 (the flat weight rows then ARE zero-padded k-tiles) and the input pitch already is.  Other shapes keep the
        fp32-storage bf16 kernel."""
        # (+ the limits of the bf16 normaliser the plans commit to, pulse_rms_normalize_b16: at least 64 columns, rows of at most 3072;
        #  other shapes keep the fp32-storage bf16 kernel instead of dying in the normaliser's argument check)
        return (self.act in (ACT_RELU, ACT_SILU) and all(uu % 32 == 0 for uu in self.units) and self.in_pitch % 32 == 0
                and 64 <= self.in_dim and self.in_pitch <= 3072 and os.environ.get("PULSE_BF16_STORAGE", "1") != "0")

    # ------------------------------------------------------------------ bf16-storage training plans (mixed_precision)
    def _plan_forward_b16(self, ws, m):
        """Actor + critic training forward over bf16 operands: x16 -> h16[l] -> heads (fp32: the loss kernel reads them)."""
        u, f, f16 = self.units, self.flat, self._flat16
        pre = ws.get("z")
        p = K.Plan()
        # the bf16 weight image and the W_l^T images of the backward pass (W_l^T (bf16) of both nets: out (2, u_{l-1}, u_l)) in one launch at
        # the head of the training forward: the weights do not change between a minibatch's forward and its backward
        trs = [dict(x=f, out=self._wt16[l], x_off=self.w_off[l], rows=u[l], cols=u[l - 1], ld_in=u[l - 1], ld_out=u[l], batch=2,
                    stride_in=u[l] * u[l - 1], stride_out=u[l - 1] * u[l]) for l in range(1, len(u))]
        if len(trs) <= 4:
            p.weights_b16(f, f16, self.n_flat, trs)
        else:
            p.refresh_b16(f, f16, self.n_flat)
            for t in trs:
                p.transpose_b16(t.pop("x"), t.pop("out"), **t)
        hm = ws.get("hmask8")
        mk = lambda l, per_net: {} if hm is None else dict(relu_mask8=hm[l], stride_mask8=u[l] // 8 if per_net else 0)
        for l, uu in enumerate(u):
            if l == 0:
                k = self.in_w[0]
                p.gemm_b16(ws["x16"], f16, M=m, N=2 * uu, K=self.in_dim, ldb=k, b_off=self.w_off[0], Cp=ws["h16"][0], bias=f, bias_off=self.b_off[0],
                           activation=self.act, C2=pre[0] if pre else None, ldc2=2 * uu, **mk(0, False))
            else:
                up = u[l - 1]
                p.gemm_b16(ws["h16"][l - 1], f16, M=m, N=uu, K=up, ldb=up, b_off=self.w_off[l], batch=2, stride_a=up, stride_b=uu * up,
                           Cp=ws["h16"][l], stride_cp=uu, bias=f, bias_off=self.b_off[l], stride_bias=uu, activation=self.act,
                           C2=pre[l] if pre else None, ldc2=2 * uu, stride_c2=uu, **mk(l, True))
        uL, ap, hr = u[-1], self.a_pitch, self.head_rows
        p.gemm_b16(ws["h16"][-1], f16, M=m, N=hr, K=uL, ldb=uL, b_off=self.wh_off, batch=2, stride_a=uL, stride_b=hr * uL, C=ws["heads"], ldc=2 * ap,
                   stride_c=ap, bias=f, bias_off=self.bh_off, stride_bias=ap, algo_n=(self.actions_num + 1) / 2.0)
        return p

    def _plan_backward_b16(self, ws, m):
        """Backward over bf16 operands.  Same order as _plan_backward (dX chain, layer-1 weight gradient, then the rest: the data-parallel
        bucket split).  Input gradients run in the FORWARD form over W^T copies rebuilt at the head of the plan; weight gradients read
        dZ and the activations in their natural row-major storage (both [red][out]) into the fp32 split-K slabs; bias gradients are
        column sums of the bf16 dZ matrices (slab 0)."""
        u, f, f16, S = self.units, self.flat, self._flat16, self.split_k
        L, uL, ap, hr, hp = len(u), u[-1], self.a_pitch, self.head_rows, ws["head_pitch16"]
        slabs, P = self._slabs, self.n_flat
        slabs.zero_()                       # the slabs a layer's weight-gradient launch does not write must read as zero
        egrad = EPI_RELU_GRAD if self.act == ACT_RELU else EPI_SILU_GRAD
        aux = ws["h16"] if self.act == ACT_RELU else ws["z"]
        ld_aux = lambda t: t.stride(0)
        dz, h16, dh16 = ws["dz16"], ws["h16"], ws["dheads16"]
        p = K.Plan()                        # (the W_l^T images were written at the head of this minibatch's forward plan)
        # heads -> dZ_L of both nets (K = head rows: tiny; W_heads read as the [red][out] operand it is)
        # every launch that produces a layer's dZ also hands over per-row-tile column sums of what it stored (out_colsum): the layer's bias
        # gradient is their ordered sum -- a few-KB reduce into slab 0 instead of a pass over the (m, 2 u) dZ matrix
        cs = ws["colsum16"] = [torch.zeros(K.gemm_x3p_row_tiles(m, uu, 2), 2 * uu, dtype=torch.float32, device=self.device) for uu in u]
        hm = ws.get("hmask8")              # [r6] the forward's sign bits: one byte per 8 activations instead of 16 bytes of h16

        def deriv(l):
            if hm is not None:
                return dict(relu_mask8=hm[l], stride_mask8=u[l] // 8)
            return dict(aux=aux[l], ldaux=ld_aux(aux[l]), stride_aux=u[l])

        p.gemm_b16(dh16, f16, M=m, N=uL, K=hr, ldb=uL, b_off=self.wh_off, b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=hp, stride_b=hr * uL,
                   Cp=dz[-1], stride_cp=uL, epilogue=egrad, algo_k=(self.actions_num + 1) / 2.0, out_colsum=cs[-1], stride_out_colsum=uL, **deriv(L - 1))
        for l in range(L - 1, 0, -1):
            uu, up = u[l], u[l - 1]
            p.gemm_b16(dz[l], self._wt16[l], M=m, N=up, K=uu, ldb=uu, batch=2, stride_a=uu, stride_b=up * uu, Cp=dz[l - 1], stride_cp=up,
                       epilogue=egrad, out_colsum=cs[l - 1], stride_out_colsum=up, **deriv(l - 1))
        uu, k = u[0], self.in_w[0]
        s1 = self._l0_slabs = ws["l0_slabs"] = K.dw_split_b16(2 * uu, k, 1, S)
        ws["layer_slabs"] = [s1] + [S] * (L - 1)
        p.gemm_b16(dz[0], ws["x16"], M=2 * uu, N=k, K=m, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, C=slabs, ldc=k, c_off=self.w_off[0],
                   split_k=s1, split_stride=P, algo_n=self.in_dim)
        p.call_partial_reduce(cs[0], cs[0].shape[0], 2 * uu, slabs, self.b_off[0])                           # bias 1 -> slab 0
        p.split = len(p.ops)
        hs, HS = self._head_scratch, self._head_split if m >= 96 * self._head_split else 1
        hb = self.bh_off - self.wh_off
        p.gemm_b16(dh16, h16[-1], M=hr, N=uL, K=m, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=hp, stride_b=uL, C=hs, ldc=uL,
                   stride_c=hr * uL, split_k=HS, split_stride=hs.stride(0), algo_k=m * (self.actions_num + 1) / (2.0 * hr))
        # head bias gradients: column sums of [d mu | pad] and [d value | pad] into the scratch rows' bias columns (b_mu: A entries, b_value: 1)
        p.colsum_b16(dh16, m, ap, 2 * hp, hs, HS, hs.stride(0), hb)
        p.colsum_b16(dh16, m, ap, 2 * hp, hs, HS, hs.stride(0), hb + ap, x_off=hp)
        p.call_partial_reduce(hs, HS, hb + 2 * ap, slabs, self.wh_off)
        for l in range(L - 1, 0, -1):
            uu, up = u[l], u[l - 1]
            sl = ws["layer_slabs"][l] = K.dw_split_b16(uu, up, 2, S)
            p.gemm_b16(dz[l], h16[l - 1], M=uu, N=up, K=m, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=uu, stride_b=up,
                       C=slabs, ldc=up, stride_c=uu * up, c_off=self.w_off[l], split_k=sl, split_stride=P)
            p.call_partial_reduce(cs[l], cs[l].shape[0], 2 * uu, slabs, self.b_off[l])
        return p

    def _plan_forward(self, ws, m, n0, cnt, bf16=False, planar=False, masks=False):
        """nets n0 .. n0+cnt-1 (0 = actor, 1 = critic).  ``masks``: the hidden layers also write their ReLU bit masks (ws['hmask'])."""
        u, f = self.units, self.flat
        pre = ws.get("z")
        p = K.Plan(bf16=bf16)
        hm = ws["hmask"] if masks else None
        mk = lambda l, net0, per_net: ({} if hm is None else
                                       dict(relu_mask=hm[l], ld_mask=hm[l].stride(0), mask_off=net0 * (u[l] // 4), stride_mask=u[l] // 4 if per_net else 0))
        for l, uu in enumerate(u):
            k = self.in_w[l]
            if l == 0 and planar:
                wp, xp = self._w1p, ws["xp"]
                # W1 of both nets is one contiguous (2 u1, in_pitch) matrix of the flat buffer: its planes are refreshed here, so no writer of
                # the parameters (optimiser step, checkpoint load, broadcast, a test poking the flat buffer) can leave them stale
                p.call("pulse_split_planes", f.data_ptr() + 4 * (self.w_off[0] + n0 * uu * k), k, cnt * uu, self.in_dim,
                       wp.data_ptr() + 2 * n0 * uu * wp.stride(1), wp.stride(0), wp.stride(1), 0, None)
                p.gemm_x3p(xp, wp, M=m, N=cnt * uu, K=self.in_dim, C=ws["h"][0], ldc=2 * uu, bias=f, activation=self.act,
                           b_off=n0 * uu * wp.stride(1), bias_off=self.b_off[0] + n0 * uu, c_off=n0 * uu,
                           C2=pre[0] if pre else None, ldc2=2 * uu, c2_off=n0 * uu)
            elif l == 0:   # both nets read the same input: one GEMM of N = cnt*u1
                p.gemm(ws["x"], f, ws["h"][0], M=m, N=cnt * uu, K=k, lda=k, ldb=k, ldc=2 * uu, bias=f, activation=self.act,
                       algo_k=self.in_dim, b_off=self.w_off[0] + n0 * uu * k, bias_off=self.b_off[0] + n0 * uu, c_off=n0 * uu,
                       C2=pre[0] if pre else None, ldc2=2 * uu, c2_off=n0 * uu, **mk(0, n0, False))
            else:
                up = u[l - 1]
                p.gemm(ws["h"][l - 1], f, ws["h"][l], M=m, N=uu, K=up, lda=2 * up, ldb=up, ldc=2 * uu, bias=f, activation=self.act,
                       batch=cnt, stride_a=up, stride_b=uu * up, stride_c=uu, stride_bias=uu,
                       a_off=n0 * up, b_off=self.w_off[l] + n0 * uu * up, bias_off=self.b_off[l] + n0 * uu, c_off=n0 * uu,
                       C2=pre[l] if pre else None, ldc2=2 * uu, stride_c2=uu, c2_off=n0 * uu, **mk(l, n0, True))
        uL, ap, hr = u[-1], self.a_pitch, self.head_rows
        # heads (batched: mu from the actor half of h_L, value from the critic half)
        p.gemm(ws["h"][-1], f, ws["heads"], M=m, N=hr, K=uL, lda=2 * uL, ldb=uL, ldc=2 * ap, bias=f, batch=cnt, stride_a=uL,
               stride_b=hr * uL, stride_c=ap, stride_bias=ap, a_off=n0 * uL, b_off=self.wh_off + n0 * hr * uL,
               bias_off=self.bh_off + n0 * ap, c_off=n0 * ap, algo_n=(self.actions_num + 1) / 2.0 if cnt == 2 else (self.actions_num if n0 == 0 else 1))
        return p

    def forward(self, ws, m):
        """Actor + critic forward on the normalised input in ws['x'] -> ws['heads'] (mu | value)."""
        if self._take_planes(ws):
            ws["plan_fwd_planar"].run()
            return
        if self.training and ws.get("b16"):
            # bf16-storage training pass: the normaliser wrote ws['x16'] directly (the caller raised ws['x16_fresh']); anyone who filled the
            # fp32 ws['x'] instead gets it rounded here
            if not ws.pop("x16_fresh", False):
                K.to_b16(ws["x"], ws["x16"])
            ws["plan_fwd_train"].run()
            return
        # (a workspace with ReLU bit masks always runs the mask-writing plan: its backward must never see the masks of an older forward)
        (ws["plan_fwd_train"] if ((self.training or "hmask" in ws) and "plan_fwd_train" in ws) else ws["plan_fwd"]).run()

    def _take_planes(self, ws):
        """True once per normaliser pass that also wrote ws['xp'] (the caller raised ws['xp_fresh']); anyone who fills ws['x'] by other
        means gets the in-kernel-split path.  The bf16 training passes (mixed_precision) do not use the planes."""
        fresh = ws.get("xp_fresh", False)
        if fresh:
            ws["xp_fresh"] = False
        return fresh and not (self.training and self.mixed_precision)

    def eval_critic(self, ws, m):
        """Critic only (CommonAgent._eval_critic, common_agent.py:551-562) -> ws['val']."""
        (ws["plan_critic_planar"] if self._take_planes(ws) else ws["plan_critic"]).run()

    # ------------------------------------------------------------------ backward
    def _plan_backward(self, ws, m):
        u, f, S = self.units, self.flat, self.split_k
        L, uL, ap, hr = len(u), u[-1], self.a_pitch, self.head_rows
        slabs, P = self._slabs, self.n_flat
        egrad = EPI_RELU_GRAD if self.act == ACT_RELU else EPI_SILU_GRAD
        aux = ws["h"] if self.act == ACT_RELU else ws["z"]
        p = K.Plan(bf16=self.mixed_precision)
        hm = ws.get("hmask")                 # ReLU bit masks the training forward wrote: the input-gradient epilogues read 1 bit, not 4 bytes, per element

        def deriv(l):
            if hm is not None:
                return dict(relu_mask=hm[l], ld_mask=hm[l].stride(0), stride_mask=u[l] // 4)
            return dict(aux=aux[l], ldaux=2 * u[l], stride_aux=u[l])

        dhd = ws["dheads"]
        # Order: the whole dX chain first, then the layer-1 weight gradient -- the largest GEMM and the largest gradient bucket
        # (flat elements < w_off[1]: 8 of 12 MB for [1024, 512]) -- then the upper layers' and the heads' weight gradients.  With
        # data parallelism the big bucket is on the wire while ~270 us of GEMMs are still to run, and only the small one is exposed
        # (the forward order of the weight-gradient GEMMs does not matter: nothing consumes them before the optimiser).
        # heads -> dH_L for both nets in one launch (activation derivative fused)
        p.gemm(dhd, f, ws["dh"][-1], M=m, N=uL, K=hr, lda=2 * ap, ldb=uL, ldc=2 * uL, b_layout=GEMM_OUT_CONTIG, batch=2,
               stride_a=ap, stride_b=hr * uL, stride_c=uL, b_off=self.wh_off, epilogue=egrad, algo_k=(self.actions_num + 1) / 2.0,
               **deriv(L - 1))
        for l in range(L - 1, 0, -1):
            uu, up = u[l], u[l - 1]
            p.gemm(ws["dh"][l], f, ws["dh"][l - 1], M=m, N=up, K=uu, lda=2 * uu, ldb=up, ldc=2 * up, b_layout=GEMM_OUT_CONTIG,
                   batch=2, stride_a=uu, stride_b=uu * up, stride_c=up, b_off=self.w_off[l], epilogue=egrad, **deriv(l - 1))
        p.split_dx = len(p.ops)             # everything before this point is the input-gradient chain; the weight gradients after it are independent of each other
        # weight gradients; the bias gradients (column sums of dz) ride along as the GEMM's per-slab row sums
        uu, k = u[0], self.in_w[0]
        s1 = self._l0_slabs = ws["l0_slabs"] = K.dw_split_x3(2 * uu, k, 1, S, K=m)
        ws["layer_slabs"] = [s1] + [S] * (L - 1)
        p.gemm(ws["dh"][0], ws["x"], slabs, M=2 * uu, N=k, K=m, lda=2 * uu, ldb=k, ldc=k, a_layout=GEMM_OUT_CONTIG,
               b_layout=GEMM_OUT_CONTIG, c_off=self.w_off[0], split_k=s1, split_stride=P, algo_n=self.in_dim,
               rowsum=slabs, rowsum_off=self.b_off[0])
        p.split = len(p.ops)                # everything after this point only touches gradient elements >= w_off[1] (layers 2.., heads)
        # head weight gradients: tiny outputs (2 x [A][uL]) over a long reduction -> split wide into a scratch
        hs, HS = self._head_scratch, self._head_split if m >= 32 * self._head_split else 1
        hb = self.bh_off - self.wh_off                       # the scratch rows mirror the flat layout [head weights | head biases]
        p.gemm(dhd, ws["h"][-1], hs, M=hr, N=uL, K=m, lda=2 * ap, ldb=2 * uL, ldc=uL, a_layout=GEMM_OUT_CONTIG,
               b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=ap, stride_b=uL, stride_c=hr * uL, split_k=HS, split_stride=hs.stride(0),
               algo_k=m * (self.actions_num + 1) / (2.0 * hr), rowsum=hs, rowsum_off=hb, stride_rowsum=ap)
        p.call_partial_reduce(hs, HS, hb + 2 * ap, slabs, self.wh_off)
        for l in range(L - 1, 0, -1):
            uu, up = u[l], u[l - 1]
            sl = K.dw_split_x3(uu, up, 2, self.n_slabs, K=m)
            ws["layer_slabs"][l] = sl
            p.gemm(ws["dh"][l], ws["h"][l - 1], slabs, M=uu, N=up, K=m, lda=2 * uu, ldb=2 * up, ldc=up, a_layout=GEMM_OUT_CONTIG,
                   b_layout=GEMM_OUT_CONTIG, batch=2, stride_a=uu, stride_b=up, stride_c=uu * up, c_off=self.w_off[l],
                   split_k=sl, split_stride=P, rowsum=slabs, rowsum_off=self.b_off[l], stride_rowsum=uu)
        return p

    supports_fused_sqnorm = True

    def _slab_regions(self, ws):
        """[(offset, count, slabs)] of the flat gradient in order: every layer's range [W_l | b_l] with the slab count ITS weight-gradient launch
        of this workspace writes (ws['layer_slabs']; they differ per layer and per minibatch size since the launcher has two tilings), then the
        heads' range, whose sums come from the plan's own partial rows (slab 0 holds them when the plan's small reduces run; the other slabs
        are never written there).  A reduce that read a fixed count everywhere would add whatever another workspace's launch left in the slabs
        this one does not write (round-4 advisor finding)."""
        L, ls = len(self.units), ws["layer_slabs"]
        ends = [self.w_off[l + 1] if l + 1 < L else self.wh_off for l in range(L)]
        base = [(self.w_off[l], ends[l] - self.w_off[l], ls[l]) for l in range(L)]
        return base + [(self.wh_off, self.n_flat - self.wh_off, 1)]

    def _build_reduce_all(self, ws, plan):
        """Regions of the whole-gradient reduce: every layer's range with the slabs its launch wrote, and -- carved out of those -- every range
        whose partials the plan registered (Plan.call_partial_reduce), summed from their own buffers.  More than 8 regions (deep MLPs): the
        plan keeps its small reduces and the ranges are reduced one by one."""
        regions, fused = K.carve_reduce_regions(self._slab_regions(ws), getattr(plan, "partial_reduces", []))
        ws["reduce_all_fused"] = fused
        return K.ReduceGrads(self._slabs, self.n_flat, regions, self.grad) if len(regions) <= 8 else None

    def backward(self, ws, m, grad_scale=1.0, on_bucket=None, sq_partials=None):
        """Given d loss/d(mu, value) in ws['dheads'], fill self.grad (flat, same layout as self.flat).
        Deterministic: split-K slabs + ordered reduces.

        ``on_bucket(grad_view)``: data-parallel overlap hook.  The plan computes the layer-1 weight gradient (flat elements < w_off[1],
        8 of 12 MB for [1024, 512]) before the upper layers' and the heads' weight gradients: that bucket is reduced and handed over first
        so its all-reduce runs beside the remaining GEMMs; the small bucket follows."""
        plan = ws["plan_bwd"]
        mode = bool(ws.get("b16"))
        if self._slab_mode is not None and self._slab_mode != mode:
            # the fp32- and bf16-storage plans write different slab counts per region: a slab the other mode wrote and this one leaves alone
            # must read as zero again (round-4 advisor finding: toggling mixed_precision on a live network)
            self._slabs.zero_()
        self._slab_mode = mode
        if on_bucket is None or len(self.units) < 2:
            # ONE reduce launch over the flat gradient: the layer-1 region reads only the slabs its weight-gradient launch wrote, the partials
            # the plan registered (bias column sums, the heads' wide split) are read where they lie -- their small reduce launches are left
            # out of the plan --, and the launch leaves the per-block sums of squares the gradient-norm clip needs (``sq_partials``)
            if "reduce_all" not in ws:
                ws["reduce_all"] = self._build_reduce_all(ws, plan)
            rg = ws["reduce_all"]
            # (round 4 / 5 measured the upper layers' weight gradients on a side stream beside the layer-1 one: +1.3 % on the 128 x 128 tiling, -1 %
            #  on the 256 x 256 one, whose launches each hold every CU's whole register file -- profiles/r05_ab_runs.txt; removed in round 6)
            if rg is not None:
                plan.run(skip_partial_reduces=ws["reduce_all_fused"])
                rg.run(scale=grad_scale, sq_partials=sq_partials)
            else:                                                # deep MLP: more ranges than one fused launch takes
                plan.run()
                for off, cnt, ns in self._slab_regions(ws):
                    K.reduce_slabs(self._slabs, ns, self.n_flat, cnt, self.grad, scale=grad_scale, slabs_off=off, out_off=off)
                if sq_partials is not None:
                    K.sqnorm_partial(self.grad, self.n_flat, sq_partials)
            if on_bucket is not None:
                on_bucket(self.grad)
            return self.grad
        cut = self.w_off[1]
        regions = self._slab_regions(ws)
        plan.run(0, plan.split)
        # the layer-1 region [0, cut) holds only the slabs its dW GEMM wrote: reduce just those
        K.reduce_slabs(self._slabs, regions[0][2], self.n_flat, cut, self.grad, scale=grad_scale)
        on_bucket(self.grad[:cut])
        plan.run(plan.split, None)
        for off, cnt, ns in regions[1:]:
            K.reduce_slabs(self._slabs, ns, self.n_flat, cnt, self.grad, scale=grad_scale, slabs_off=off, out_off=off)
        on_bucket(self.grad[cut:])
        return self.grad
