"""AMP discriminator on the gfx950 kernels, including the gradient penalty's double backward by hand.

Mirrors AMPBuilder.Network._build_disc / eval_disc (phc/learning/amp_network_builder.py:213-249) and
AMPAgent._disc_loss (phc/learning/amp_agent.py:895-952):

    D(x) = w3 . relu(W2 relu(W1 x + b1) + b2) + b3
    disc_loss = 0.5 (BCE(D(agent U replay), 0) + BCE(D(demo), 1))
              + disc_logit_reg * ||w3||^2 + disc_grad_penalty * mean_demo || dD/dx ||^2
              + disc_weight_decay * (||W1||^2 + ||W2||^2 + ||w3||^2)

The reference gets the penalty's parameter gradient from autograd with create_graph=True.  For a ReLU MLP the
input gradient is itself a (mask-gated) linear chain,   dD/dx = W1^T (m1 * (W2^T (m2 * w3))),   so its
parameter gradients are ordinary GEMMs (masks are piecewise constant):
    u2 = m2 * w3,  u1 = m1 * (u2 W2),  g = u1 W1,  P = mean ||g||^2,  dg = 2 c g / B
    dW1 += u1^T dg ;  dt1 = m1 * (dg W1^T) ;  dW2 += u2^T dt1 ;  dw3 += sum_B m2 * (dt1 W2^T)
Both gradient paths of a weight are accumulated by STACKING them along the reduction (batch) dimension:
rows [0, 3B) of the operand buffers hold the BCE path (dz, activations), rows [3B, 4B) the penalty path
(u, d-terms), and one batch-split GEMM per weight reduces over all 4B rows.  Every product above is one of
the three layouts of gemm_f32_kernel with the ReLU-mask epilogue; no autograd graph, no second-order tape.

mixed_precision (the reference's autocast around calc_gradients, amp_agent.py:671): the same products on bf16 STORAGE -- X, H, Z and
the logit gradients are bf16 matrices, the GEMMs are pulse_gemm_x3p(planes = 1), input-gradient products read W^T copies rebuilt once
per pass so they run in the forward form, weight gradients go to the fp32 split-K slabs (``_plans_b16``).
"""
import os

import torch

from .. import kernels as K
from .._lib import ACT_NONE, ACT_RELU, EPI_RELU_GRAD, GEMM_OUT_CONTIG
from . import network as _network
from .graph import Linear, ParamBook, init_linear_, r4


def r32(x):
    return (x + 31) // 32 * 32


class DiscNetwork:
    def __init__(self, params, amp_input_dim, device="cuda:0", split_k=8):
        self.device = torch.device(device)
        units = [int(u) for u in params["disc"]["units"]]
        if len(units) != 2 or params["disc"]["activation"] != "relu":
            raise NotImplementedError("discriminator: two ReLU hidden layers (every shipped config)")
        self.u1, self.u2 = units
        self.k0 = int(amp_input_dim)
        self.k0p = r32(self.k0)          # whole 32-deep k-tiles: X rows and W1 rows are zero-padded to it (what the bf16-storage GEMM reads)
        self.book = ParamBook(self.device, split_k)
        self.l1 = Linear(self.book, "a2c_network._disc_mlp.0", self.k0, self.u1, ACT_RELU, pitch_align=32)
        self.l2 = Linear(self.book, "a2c_network._disc_mlp.2", self.u1, self.u2, ACT_RELU)
        self.l3 = Linear(self.book, "a2c_network._disc_logits", self.u2, 1, ACT_NONE)
        self.book.finalize()
        self.flat, self.grad, self.n_flat = self.book.flat, self.book.grad, self.book.n_flat
        self._ws = {}
        self.mixed_precision = False      # True: the training passes (forward on the 3b rows, BCE / penalty backward, weight gradients) on the bf16 MFMA
        self._flat16 = self._w1t16 = self._w2t16 = None
        self._reg_partials = torch.zeros(1024, 8, device=self.device)
        self.reset_parameters()

    def b16_storage_ok(self):
        # hidden widths in whole 32-deep k-tiles, and the limits of the bf16 normaliser the plans commit to (pulse_rms_normalize_b16: at least 64
        # columns, rows of at most 3072): other shapes keep the fp32-storage bf16 kernel instead of dying in the normaliser's argument check
        return (self.u1 % 32 == 0 and self.u2 % 32 == 0 and 64 <= self.k0 and self.k0p <= 3072
                and os.environ.get("PULSE_BF16_STORAGE", "1") != "0")

    def reset_parameters(self, generator=None):
        init_linear_(self.book, self.l1, generator)
        init_linear_(self.book, self.l2, generator)
        # torch.nn.init.uniform_(_disc_logits.weight, -1, 1); zero bias (amp_network_builder.py:246-247)
        self.book.set(self.l3.w.name, torch.empty(1, self.u2).uniform_(-1.0, 1.0, generator=generator))
        self.book.set(self.l3.b.name, torch.zeros(1))

    # ---- reference-named parameters
    def state_dict(self, buf=None):
        out = {}
        for lin in (self.l1, self.l2, self.l3):
            out[lin.w.name] = self.book.get(lin.w.name, buf).clone()
            out[lin.b.name] = self.book.get(lin.b.name, buf).reshape(-1).clone()
        return out

    def gradients(self):
        return self.state_dict(self.book.grad)

    def load_state_dict(self, sd, strict=True):
        for lin in (self.l1, self.l2, self.l3):
            for p in (lin.w, lin.b):
                if p.name in sd:
                    self.book.set(p.name, sd[p.name])
                elif strict:
                    raise KeyError(p.name)

    def get_disc_logit_weights(self):
        return self.book.get(self.l3.w.name).reshape(-1)

    # ---- workspaces: b rows per stream (agent / replay / demo), 4b stacked rows
    def workspace(self, b):
        key = (b, bool(self.mixed_precision))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev = self.device
        z = lambda r, c: torch.zeros(r, c, dtype=torch.float32, device=dev)
        m = 4 * b
        ws = {"b": b, "L": z(m, 4), "G": z(b, self.k0p), "pen_partials": z(1, 256).view(-1), "b16": bool(self.mixed_precision and self.b16_storage_ok())}
        ws["logits"] = ws["L"][:3 * b, :1]
        if ws["b16"]:
            i16 = lambda r, c: torch.zeros(r, c, dtype=torch.int16, device=dev)
            ws.update({"X": i16(m, self.k0p), "H1": i16(m, self.u1), "H2": i16(m, self.u2), "Z1": i16(m, self.u1), "Z2": i16(m, self.u2),
                       "dL16": i16(m, 32)})
            ws["dL16"][3 * b:, 0] = 0x3F80                  # bf16(1.0): the penalty path's "ones" column for dw3
            if self._flat16 is None:
                self._flat16 = i16(1, (self.n_flat + 7) // 8 * 8).view(-1)
                self._w1t16 = i16(self.k0, self.u1)          # W1^T: dD/dx = u1 W1 in the forward form
                self._w2t16 = i16(self.u1, self.u2)          # W2^T
            if _network.RELU_BITMASK:   # [r6] sign bits of H1 / H2 (rows of the 3b forward rows): what the six relu-grad launches read instead of the activations
                ws["M1"], ws["M2"] = K.alloc_relu_mask8(m, self.u1, dev), K.alloc_relu_mask8(m, self.u2, dev)
            ws["fwd"], ws["bwd_bce"], ws["pen_fwd"], ws["pen_bwd"], ws["wgrad"] = self._plans_b16(ws)
        else:
            ws.update({"X": z(m, self.k0p), "H1": z(m, self.u1), "H2": z(m, self.u2), "Z1": z(m, self.u1), "Z2": z(m, self.u2), "dL": z(m, 4)})
            ws["dL"][3 * b:, 0] = 1.0                      # the penalty path's "ones" column for dw3
            ws["dlogits"] = ws["dL"][:3 * b, :1]
            ws["fwd"] = self._plan_forward(ws, 3 * b, bf16=self.mixed_precision)
            ws["bwd_bce"], ws["pen_fwd"], ws["pen_bwd"], ws["wgrad"] = self._plans_backward(ws)
        self._ws[key] = ws
        return ws

    def _plans_b16(self, ws):
        """The five plans of a training pass over bf16 operands (module docstring): forward on the 3b rows, BCE backward, penalty forward
        (demo rows), penalty backward, weight / bias gradients over the 4b stacked rows."""
        f, f16, b = self.flat, self._flat16, ws["b"]
        u1, u2, k0, k0p = self.u1, self.u2, self.k0, self.k0p
        S, P, slabs = self.book.split_k, self.book.n_flat, self.book.slabs
        w1, w2, w3 = self.l1.w, self.l2.w, self.l3.w
        X, H1, H2, Z1, Z2, dL = ws["X"], ws["H1"], ws["H2"], ws["Z1"], ws["Z2"], ws["dL16"]
        r3, demo = 3 * b, 2 * b
        slabs.zero_()                          # the slabs a weight-gradient launch does not write must read as zero
        fwd = K.Plan()
        # the bf16 weight image and the two W^T images of the backward passes in one launch (the weights do not change inside a minibatch)
        fwd.weights_b16(f, f16, self.n_flat, [dict(x=f, out=self._w2t16, x_off=w2.off, rows=u2, cols=u1, ld_in=w2.pitch, ld_out=u2),
                                              dict(x=f, out=self._w1t16, x_off=w1.off, rows=u1, cols=k0, ld_in=w1.pitch, ld_out=u1)])
        M1, M2 = ws.get("M1"), ws.get("M2")
        # the derivative of ReLU for rows [row0, ..) of H1 / H2: the forward's sign bits when the workspace has them, else the activations themselves
        d1 = lambda row0: (dict(relu_mask8=M1, mask8_off=row0 * M1.stride(0)) if M1 is not None else dict(aux=H1, ldaux=u1, aux_off=row0 * u1))
        d2 = lambda row0: (dict(relu_mask8=M2, mask8_off=row0 * M2.stride(0)) if M2 is not None else dict(aux=H2, ldaux=u2, aux_off=row0 * u2))
        fwd.gemm_b16(X, f16, M=r3, N=u1, K=k0, ldb=w1.pitch, b_off=w1.off, Cp=H1, bias=f, bias_off=self.l1.b.off, activation=ACT_RELU,
                     **({} if M1 is None else dict(relu_mask8=M1)))
        fwd.gemm_b16(H1, f16, M=r3, N=u2, K=u1, ldb=w2.pitch, b_off=w2.off, Cp=H2, bias=f, bias_off=self.l2.b.off, activation=ACT_RELU,
                     **({} if M2 is None else dict(relu_mask8=M2)))
        fwd.gemm_b16(H2, f16, M=r3, N=1, K=u2, ldb=w3.pitch, b_off=w3.off, C=ws["L"], ldc=4, bias=f, bias_off=self.l3.b.off)
        # (1) BCE path over the 3b forward rows: dz2 = (dL w3) * m2 ; dz1 = (dz2 W2) * m1 -- the latter in the forward form over W2^T
        bce = K.Plan()
        # (both launches also hand over the column sums of the dZ rows they store: the bias gradients of layers 2 and 1, see the reduces below)
        cs2 = ws["colsum2"] = torch.zeros(K.gemm_x3p_row_tiles(r3, u2, 1), u2, dtype=torch.float32, device=self.device)
        cs1 = ws["colsum1"] = torch.zeros(K.gemm_x3p_row_tiles(r3, u1, 1), u1, dtype=torch.float32, device=self.device)
        bce.gemm_b16(dL, f16, M=r3, N=u2, K=1, ldb=w3.pitch, b_off=w3.off, b_layout=GEMM_OUT_CONTIG, Cp=Z2, epilogue=EPI_RELU_GRAD, out_colsum=cs2, **d2(0))
        bce.gemm_b16(Z2, self._w2t16, M=r3, N=u1, K=u2, Cp=Z1, epilogue=EPI_RELU_GRAD, out_colsum=cs1, **d1(0))
        # (2a) penalty forward on the demo rows (stacked as rows 3b..): u2 = m2 * w3, u1 = m1 * (u2 W2), g = u1 W1 (fp32: its square sum is the penalty)
        pf = K.Plan()
        pf.gemm_b16(dL, f16, M=b, N=u2, K=1, a_off=r3 * 32, ldb=w3.pitch, b_off=w3.off, b_layout=GEMM_OUT_CONTIG, Cp=Z2, cp_off=r3 * u2,
                    epilogue=EPI_RELU_GRAD, **d2(demo))
        pf.gemm_b16(Z2, self._w2t16, M=b, N=u1, K=u2, a_off=r3 * u2, Cp=Z1, cp_off=r3 * u1, epilogue=EPI_RELU_GRAD, **d1(demo))
        pf.gemm_b16(Z1, self._w1t16, M=b, N=k0, K=u1, a_off=r3 * u1, C=ws["G"], ldc=k0p)
        # (2b) penalty backward: dt1 = (dg W1^T) * m1 -> H1[3b:] ;  m2 * (dt1 W2^T) -> H2[3b:]      (dg lives in X[3b:]); W as stored = forward form
        pb = K.Plan()
        pb.gemm_b16(X, f16, M=b, N=u1, K=k0, a_off=r3 * k0p, ldb=w1.pitch, b_off=w1.off, Cp=H1, cp_off=r3 * u1, epilogue=EPI_RELU_GRAD, **d1(demo))
        pb.gemm_b16(H1, f16, M=b, N=u2, K=u1, a_off=r3 * u1, ldb=w2.pitch, b_off=w2.off, Cp=H2, cp_off=r3 * u2, epilogue=EPI_RELU_GRAD, **d2(demo))
        # (3) weight gradients over all 4b stacked rows (fp32 slabs), bias gradients = column sums over the 3b BCE rows (slab 0)
        wg = K.Plan()
        m = 4 * b
        t256 = lambda mm, nn: ((mm + 255) // 256) * ((nn + 127) // 128)
        ws["w_slabs"] = (K.dw_split_b16(u1, k0, 1, S), 1, S)          # slabs written for W1 / W2 (summed into slab 0 by its own reduce) / w3
        wg.gemm_b16(Z1, X, M=u1, N=k0, K=m, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, C=slabs, ldc=w1.pitch, c_off=w1.off,
                    split_k=ws["w_slabs"][0], split_stride=P)
        # the two small outputs would leave most of the chip idle at S splits (16 and 4 tiles of 256 x 128): they are split wider into a
        # scratch whose ordered sum lands in slab 0 (slabs 1.. of these regions stay zero)
        def wide(A, B, M_, N_, lin, lda_note=None):
            tiles = t256(M_, N_)
            split = 1
            while tiles * split < 256 and split < 64 and m // (2 * split) >= 96:
                split *= 2
            count = lin.rows * lin.pitch
            scr = torch.zeros(split, r4(count), dtype=torch.float32, device=self.device)
            ws.setdefault("_w_scratch", []).append(scr)
            wg.gemm_b16(A, B, M=M_, N=N_, K=m, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG, C=scr, ldc=lin.pitch, split_k=split,
                        split_stride=scr.stride(0))
            wg.call_partial_reduce(scr, split, count, slabs, lin.off)
        wide(Z2, H1, u2, u1, w2)
        # d w3 = sum over the 4b stacked rows of dL[m] * H2[m][:]: a weighted column sum (no 1 x u2 GEMM), 32 row chunks into a scratch
        # (512 workgroups; one partial row per gradient slab would be 128) whose ordered sum lands in slab 0
        w3c = 32 if m >= 32 * 64 else 1
        w3s = torch.zeros(w3c, r4(u2), dtype=torch.float32, device=self.device)
        ws.setdefault("_w_scratch", []).append(w3s)
        wg.colsum_weighted_b16(H2, dL, 32, m, u2, u2, w3s, w3c, w3s.stride(0), 0)
        wg.call_partial_reduce(w3s, w3c, u2, slabs, w3.off)
        ws["w_slabs"] = (ws["w_slabs"][0], 1, 1)
        # bias gradients over the 3b BCE rows: layers 1 / 2 from the column sums their dZ launches left (slab 0), the logit bias from dL itself
        wg.call_partial_reduce(cs1, cs1.shape[0], u1, slabs, self.l1.b.off)
        wg.call_partial_reduce(cs2, cs2.shape[0], u2, slabs, self.l2.b.off)
        # (the logit bias' gradient -- the sum of dL over the 3b rows -- is left in slab 0 by the loss head itself: loss_head_b16)
        return fwd, bce, pf, pb, wg

    def _plan_forward(self, ws, m, x=None, logits=None, bf16=False):
        f = self.flat
        p = K.Plan(bf16=bf16)
        x = ws["X"] if x is None else x
        p.gemm(x, f, ws["H1"], M=m, N=self.u1, K=self.k0, lda=x.stride(0), ldb=self.l1.w.pitch, ldc=self.u1, bias=f, activation=ACT_RELU,
               b_off=self.l1.w.off, bias_off=self.l1.b.off)
        p.gemm(ws["H1"], f, ws["H2"], M=m, N=self.u2, K=self.u1, lda=self.u1, ldb=self.l2.w.pitch, ldc=self.u2, bias=f, activation=ACT_RELU,
               b_off=self.l2.w.off, bias_off=self.l2.b.off)
        p.gemm(ws["H2"], f, ws["L"], M=m, N=1, K=self.u2, lda=self.u2, ldb=self.l3.w.pitch, ldc=4, bias=f, b_off=self.l3.w.off,
               bias_off=self.l3.b.off)
        return p

    def _plans_backward(self, ws):
        f, b = self.flat, ws["b"]
        u1, u2, k0, k0p = self.u1, self.u2, self.k0, self.k0p
        S, P, slabs = self.book.split_k, self.book.n_flat, self.book.slabs
        w1, w2, w3 = self.l1.w, self.l2.w, self.l3.w
        r3, demo = 3 * b, 2 * b                                     # first penalty row / first demo row
        # (1) BCE path: dz2 = (dL w3) * m2 ; dz1 = (dz2 W2) * m1   over the 3b forward rows
        bce = K.Plan(bf16=self.mixed_precision)
        bce.gemm(ws["dL"], f, ws["Z2"], M=r3, N=u2, K=1, lda=4, ldb=w3.pitch, ldc=u2, b_layout=GEMM_OUT_CONTIG, b_off=w3.off,
                 epilogue=EPI_RELU_GRAD, aux=ws["H2"], ldaux=u2)
        bce.gemm(ws["Z2"], f, ws["Z1"], M=r3, N=u1, K=u2, lda=u2, ldb=w2.pitch, ldc=u1, b_layout=GEMM_OUT_CONTIG, b_off=w2.off,
                 epilogue=EPI_RELU_GRAD, aux=ws["H1"], ldaux=u1)
        # (2a) penalty forward on the demo rows: u2, u1, g = dD/dx
        pf = K.Plan(bf16=self.mixed_precision)
        pf.gemm(ws["dL"], f, ws["Z2"], M=b, N=u2, K=1, lda=4, ldb=w3.pitch, ldc=u2, b_layout=GEMM_OUT_CONTIG, a_off=r3 * 4, b_off=w3.off,
                c_off=r3 * u2, epilogue=EPI_RELU_GRAD, aux=ws["H2"], ldaux=u2, aux_off=demo * u2)
        pf.gemm(ws["Z2"], f, ws["Z1"], M=b, N=u1, K=u2, lda=u2, ldb=w2.pitch, ldc=u1, b_layout=GEMM_OUT_CONTIG, a_off=r3 * u2, b_off=w2.off,
                c_off=r3 * u1, epilogue=EPI_RELU_GRAD, aux=ws["H1"], ldaux=u1, aux_off=demo * u1)
        pf.gemm(ws["Z1"], f, ws["G"], M=b, N=k0, K=u1, lda=u1, ldb=w1.pitch, ldc=k0p, b_layout=GEMM_OUT_CONTIG, a_off=r3 * u1, b_off=w1.off)
        # (2b) penalty backward: dt1 = (dg W1^T) * m1 -> H1[3b:] ;  m2 * (dt1 W2^T) -> H2[3b:]      (dg lives in X[3b:])
        pb = K.Plan(bf16=self.mixed_precision)
        pb.gemm(ws["X"], f, ws["H1"], M=b, N=u1, K=k0, lda=k0p, ldb=w1.pitch, ldc=u1, a_off=r3 * k0p, b_off=w1.off, c_off=r3 * u1,
                epilogue=EPI_RELU_GRAD, aux=ws["H1"], ldaux=u1, aux_off=demo * u1)
        pb.gemm(ws["H1"], f, ws["H2"], M=b, N=u2, K=u1, lda=u1, ldb=w2.pitch, ldc=u2, a_off=r3 * u1, b_off=w2.off, c_off=r3 * u2,
                epilogue=EPI_RELU_GRAD, aux=ws["H2"], ldaux=u2, aux_off=demo * u2)
        # (3) weight gradients over all 4b stacked rows, bias gradients over the 3b BCE rows
        ws["w_slabs"] = (S, S, S)
        wg = K.Plan(bf16=self.mixed_precision)
        m = 4 * b
        wg.gemm(ws["Z1"], ws["X"], slabs, M=u1, N=k0, K=m, lda=u1, ldb=k0p, ldc=w1.pitch, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                c_off=w1.off, split_k=S, split_stride=P)
        wg.gemm(ws["Z2"], ws["H1"], slabs, M=u2, N=u1, K=m, lda=u2, ldb=u1, ldc=w2.pitch, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                c_off=w2.off, split_k=S, split_stride=P)
        wg.gemm(ws["dL"], ws["H2"], slabs, M=1, N=u2, K=m, lda=4, ldb=u2, ldc=w3.pitch, a_layout=GEMM_OUT_CONTIG, b_layout=GEMM_OUT_CONTIG,
                c_off=w3.off, split_k=S, split_stride=P)
        sc = torch.zeros(64, max(u1, u2) + 8, dtype=torch.float32, device=self.device)
        ws["_bias_scratch"] = sc
        chunks = 64 if r3 >= 1024 else 1
        for buf, n, ld, off in ((ws["Z1"], u1, u1, self.l1.b.off), (ws["Z2"], u2, u2, self.l2.b.off), (ws["dL"], 1, 4, self.l3.b.off)):
            wg.call("pulse_colsum_partial", buf.data_ptr(), r3, n, ld, chunks, sc.data_ptr(), sc.stride(0))
            wg.call("pulse_reduce_slabs", sc.data_ptr(), chunks, sc.stride(0), n, slabs.data_ptr() + 4 * off, 1.0)
        return bce, pf, pb, wg

    # ---- API ------------------------------------------------------------------------------------------
    def eval_disc(self, amp_obs_norm, out=None):
        """Inference on (m, k0p)-pitched normalised AMP observations (rollout reward): returns logits (m, 1)."""
        m = amp_obs_norm.shape[0]
        key = ("eval", m, amp_obs_norm.data_ptr())
        ws = self._ws.get(key)
        if ws is None:
            z = lambda r, c: torch.zeros(r, c, dtype=torch.float32, device=self.device)
            ws = {"X": amp_obs_norm, "H1": z(m, self.u1), "H2": z(m, self.u2), "L": z(m, 4)}
            ws["plan"] = self._plan_forward(ws, m, x=amp_obs_norm)
            self._ws[key] = ws
        ws["plan"].run()
        return ws["L"][:, :1]

    def loss_head_b16(self, ws, logits, b, scale, stats):
        """bf16-storage training pass: prediction loss statistics, d loss / d logit into ws['dL16'] AND the logit bias' gradient (their sum) into
        slab 0 of the gradient slabs -- the other slabs' entries of that range are never written and stay zero, so backward()'s reduce needs no
        column-sum launch for it.  backward() relies on this having run on the same ws."""
        K.disc_head_b16(logits, b, scale, ws["dL16"], stats, bias_grad=self.book.slabs[0, self.l3.b.off:self.l3.b.off + 1])

    def forward(self, ws):
        """Logits of the 3b stacked rows [agent | replay | demo] already normalised into ws['X'][:3b]."""
        ws["fwd"].run()
        return ws["logits"]

    def backward(self, ws, grad_penalty_coef, logit_reg, weight_decay, scale=1.0, stats=None, sq_partials=None):
        """ws['dlogits'] (fp32 storage) / ws['dL16'] (bf16 storage) holds d(total loss)/d logit for the 3b rows.  Adds the gradient penalty,
        logit regulariser and weight decay (each times ``scale`` = disc_coef) and leaves the flat gradient in self.grad.
        Returns the penalty value mean_demo ||dD/dx||^2 (device scalar), or None when ``stats`` -- a (9,) float tensor -- takes the raw
        numbers instead: [sum ||dD/dx||^2 over the demo rows, then the eight per-region sums of squared parameters of the reduce launch:
        ||W1||^2 at [1], ||W2||^2 at [3], ||w3||^2 at [5]].  ``sq_partials`` (SQ_BLOCKS = 1024 floats, one per block of the reduce launch): per-block sums of
        squares of the finished gradient."""
        if sq_partials is not None and sq_partials.numel() != self._reg_partials.shape[0]:
            raise ValueError(f"DiscNetwork.backward: sq_partials must hold {self._reg_partials.shape[0]} floats (one per reduce block), got {sq_partials.numel()}")
        b = ws["b"]
        ws["bwd_bce"].run()
        ws["pen_fwd"].run()
        # mean_demo ||dD/dx||^2 and dg = 2 c g / b (the seed of the penalty's backward pass, rows 3b.. of X) in one launch; G's pad columns
        # are never written and stay zero
        X = ws["X"]
        c = 2.0 * grad_penalty_coef * scale / b
        if ws["b16"]:
            K.disc_penalty(ws["G"], b, self.k0p, c, ws["pen_partials"], out16=X, out16_off=3 * b * self.k0p, ld16=self.k0p)
        else:
            K.disc_penalty(ws["G"], b, self.k0p, c, ws["pen_partials"], out32=X, out32_off=3 * b * self.k0p, ld32=self.k0p)
        ws["pen_bwd"].run()
        # ONE launch: split-K slab reduce (each weight region over the slabs its launch wrote), logit regulariser + weight decay gradients
        # (grad += 2 coef scale W), the three ||W||^2 of the reported loss and the sums of squares of the finished gradient for the norm clip.
        # Regions whose partials the plan registered (bias column sums, the wide-split W2, the weighted column sum of w3) are summed from
        # their own buffers: their small reduce launches are left out of the plan.
        rg = ws.get("reduce_all")
        S = self.book.split_k
        if rg is None:
            n1, n2, n3 = ws["w_slabs"]
            plain = []
            for lin, ns in ((self.l1, n1), (self.l2, n2), (self.l3, n3)):
                plain.append((lin.w.off, lin.w.rows * lin.w.pitch, ns, 0.0))
                plain.append((lin.b.off, lin.b.rows * lin.b.pitch, S, 0.0))
            assert plain[-1][0] + plain[-1][1] == self.n_flat and all(a[0] + a[1] == b[0] for a, b in zip(plain, plain[1:]))
            parts = {r[0]: r for r in getattr(ws["wgrad"], "partial_reduces", [])}
            regions = []
            for off, cnt, nsl, al0 in plain:
                pr = parts.pop(off, None)
                if pr is not None and pr[1] == cnt and cnt % 4 == 0 and pr[3].stride(0) % 4 == 0:
                    regions.append((off, cnt, pr[2], al0, pr[3], pr[3].stride(0)))
                else:
                    regions.append((off, cnt, nsl, al0))
                    if pr is not None:
                        parts[off] = pr                               # registered but not usable as a region source
            # fused only if EVERY registered partial became a region source (otherwise the plan keeps all its small reduces)
            ws["reduce_all_fused"] = any(len(r) > 4 for r in regions) and not parts
            if not ws["reduce_all_fused"]:
                regions = plain
            rg = ws["reduce_all"] = K.ReduceGrads(self.book.slabs, self.book.n_flat, regions, self.grad, flat=self.flat)
        ws["wgrad"].run(skip_partial_reduces=ws["reduce_all_fused"])
        al = [2.0 * scale * weight_decay, 0.0, 2.0 * scale * weight_decay, 0.0, 2.0 * scale * (weight_decay + logit_reg), 0.0]
        rg.run(alphas=al, sq_partials=sq_partials, w2_partials=self._reg_partials)
        if stats is not None:                                       # [sum ||dD/dx||^2 | per-region sums of squares: W1 at [1], W2 at [3], w3 at [5]]
            torch.sum(ws["pen_partials"].view(1, -1), dim=1, out=stats[0:1])
            torch.sum(self._reg_partials, dim=0, out=stats[1:9])
            return None
        return ws["pen_partials"].sum() / b
