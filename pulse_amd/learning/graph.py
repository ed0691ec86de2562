"""A small explicit-backward executor for DAGs of Linear(+activation) layers on the gfx950 GEMM kernel.

The flagship actor/critic pair has a hand-laid-out network (network.py).  The PULSE VAE policy
(phc/learning/amp_network_z_builder.py: encoder, learned prior, decoder, critic) and the AMP
discriminator (amp_network_builder.py:213-249) are wider graphs -- several MLP stacks whose inputs are
column slices / concatenations of other outputs -- so they are described as a list of ``Linear`` ops
over named, pitched activation buffers and executed as pre-built launch plans:

  forward   one GEMM per Linear, bias + ReLU/SiLU fused (SiLU also stores the pre-activation);
  backward  per Linear, in reverse order: bias gradient (column sums), weight gradient (batch-split
            GEMM into gradient slabs) and, for the column ranges that need it, the input gradient GEMM with
            the PRODUCER's activation derivative fused into its epilogue.

"Concatenation" is by construction: producers write into 4-float-aligned column segments of the
consumer's input buffer and the consumer's weight matrix is stored with the same physical column
layout (``colmap``), so there are no cat / split copies.  Parameters of a graph live in one flat
buffer (``ParamBook``): one slab reduce, one clip+Adam launch, one all-reduce.

Tiny head-level algebra (re-parameterisation, KL, log-likelihoods) is left to the caller, who seeds
``grad`` buffers and reads them back; everything heavy stays in the HIP kernels.
"""
import math

import torch

from .. import kernels as K
from .._lib import (ACT_NONE, ACT_RELU, ACT_SILU, ACT_SILU_D, EPI_BIAS_ACT, EPI_MUL_AUX, EPI_RELU_GRAD, EPI_SILU_GRAD, GEMM_OUT_CONTIG)


import os

REGION_REDUCE = os.environ.get("PULSE_BOOK_REGION_REDUCE", "1") != "0"     # ParamBook.reduce_grads region by region (0: zero fill + uniform reduce, A/B)
SILU_DERIV = os.environ.get("PULSE_SILU_DERIV", "1") != "0"       # SiLU layers keep their derivative instead of their pre-activation (forward_plan)


def r4(x):
    return (x + 3) // 4 * 4


class Param:
    def __init__(self, name, rows, cols, colmap, off, pitch):
        self.name, self.rows, self.cols, self.colmap, self.off, self.pitch = name, rows, cols, colmap, off, pitch


class ParamBook:
    """Flat parameter / gradient storage with reference-named logical views."""

    def __init__(self, device, split_k=8):
        self.device = torch.device(device)
        self.split_k = split_k
        self.params = {}
        self._n = 0
        self.flat = None

    def add(self, name, rows, cols, colmap=None, pitch_align=4):
        """rows x cols logical matrix (cols = 1 row vector for biases when rows == 1).  ``colmap``: list
        of physical column indices of the logical columns (default identity); physical pitch = roundup(max+1, pitch_align) (the
        bf16-storage GEMM wants reduction-contiguous rows that are whole zero-padded 32-deep k-tiles: pitch_align 32)."""
        phys = (max(colmap) + 1) if colmap is not None else cols
        pitch = (phys + pitch_align - 1) // pitch_align * pitch_align
        p = Param(name, rows, cols, list(colmap) if colmap is not None else None, self._n, pitch)
        self.params[name] = p
        self._n += rows * pitch
        return p

    def finalize(self, trainable=True):
        """``trainable=False``: a frozen network (teacher / decoder-in-env) only needs the parameter buffer."""
        self.n_flat = r4(self._n)
        z = lambda: torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
        if not trainable:
            self.flat = z()
            self.grad = self.exp_avg = self.exp_avg_sq = self.slabs = None
            return self
        self.flat, self.grad, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
        self.slabs = torch.zeros(self.split_k, self.n_flat, dtype=torch.float32, device=self.device)
        self.sq_partials = torch.zeros(256, device=self.device)
        self.grad_norm = torch.zeros(1, device=self.device)
        self.step = 0
        return self

    def phys(self, name, buf=None):
        p = self.params[name]
        buf = self.flat if buf is None else buf
        return buf[p.off:p.off + p.rows * p.pitch].view(p.rows, p.pitch)

    def get(self, name, buf=None):
        p = self.params[name]
        v = self.phys(name, buf)
        if p.colmap is not None:
            return v[:, p.colmap]
        return v[:, :p.cols]

    def set(self, name, value):
        p = self.params[name]
        value = value.to(self.device, torch.float32).reshape(p.rows, p.cols)
        v = self.phys(name)
        if p.colmap is not None:
            v[:, p.colmap] = value
        else:
            v[:, :p.cols] = value

    def note_slab_layout(self, name, nslabs, m):
        """Record / check the number of gradient slabs the weight-gradient launch of parameter ``name`` writes (see backward_plan)."""
        lay = self.__dict__.setdefault("_slab_layout", {})
        seen = lay.get(name)
        if seen is None:
            lay[name] = (nslabs, m)
        elif seen[0] != nslabs:
            raise RuntimeError(f"ParamBook: the weight-gradient launch of {name} writes {nslabs} gradient slab(s) at batch {m} but {seen[0]} at batch "
                               f"{seen[1]}: two training workspaces with different slab layouts would leave stale partial sums in the slabs the "
                               "other one skips.  Use one training batch size per model (or a separate ParamBook per batch size).")

    def zero_slab_ranges(self, ranges):
        """Zero the gradient slabs of the flat ranges [(lo, hi), ...] (all S slabs): the sub-networks a backward pass does NOT visit, whose
        slabs still hold another pass's gradients.  The visited ones are overwritten by their weight-gradient GEMMs; zeroing all S x n_flat
        floats per minibatch instead was 1.3 % of the cfg3 epoch."""
        for lo, hi in ranges:
            if hi > lo:
                self.slabs[:, lo:hi].zero_()

    def reduce_regions(self, untouched=()):
        """[(offset, count, nslabs)] covering [0, n_flat): per parameter the number of slabs its weight-gradient launch writes
        (note_slab_layout; split_k for a parameter no launch registered), 0 for a parameter inside one of the ``untouched`` ranges
        [(lo, hi), ...] (no launch of this pass writes it: its gradient is zero); adjacent parameters with equal counts are merged."""
        lay = self.__dict__.get("_slab_layout", {})
        regions = []
        pos = 0
        for p in sorted(self.params.values(), key=lambda q: q.off):
            size = p.rows * p.pitch
            ns = lay[p.name][0] if p.name in lay else self.split_k
            if any(lo <= p.off and p.off + size <= hi for lo, hi in untouched):
                ns = 0
            if p.off != pos:
                raise RuntimeError("ParamBook: parameters are not contiguous in the flat buffer")
            if regions and regions[-1][2] == ns:
                regions[-1] = (regions[-1][0], regions[-1][1] + size, ns)
            else:
                regions.append((p.off, size, ns))
            pos = p.off + size
        if pos < self.n_flat:                                   # the flat buffer's tail padding
            if regions and regions[-1][2] == 0:
                regions[-1] = (regions[-1][0], regions[-1][1] + self.n_flat - pos, 0)
            else:
                regions.append((pos, self.n_flat - pos, 0))
        return regions

    def reduce_grads(self, scale=1.0, untouched=None, sq_partials=None):
        """grad = scale * (sum of the slabs).  ``untouched`` (ranges of the sub-networks the pass did not visit, ComputeGraph.untouched_ranges):
        the reduce goes region by region with each parameter's OWN slab count and writes zeros over the untouched ranges [r6] -- nothing has
        to be zero-filled before the pass and no slab a launch never writes is read (the uniform form read split_k slabs of every parameter:
        930 MB per minibatch of cfg3, plus a 230 MB zero fill of the critic's slabs).  ``sq_partials`` (with ``untouched``): the launch also
        leaves the per-block sums of squares of the result (clip_grad_norm_ without its own pass).  Returns True when sq_partials were written."""
        if untouched is not None and REGION_REDUCE:
            key = tuple(untouched)
            cache = self.__dict__.setdefault("_reduce_cache", {})
            rg = cache.get(key)
            if rg is None:
                regions = self.reduce_regions(untouched)
                rg = cache[key] = K.ReduceGrads(self.slabs, self.n_flat, [(o, c, n, 0.0) for o, c, n in regions], self.grad) if len(regions) <= 32 else False
            if rg:
                rg.run(scale=scale, sq_partials=sq_partials)
                return sq_partials is not None
        if untouched is not None:
            self.zero_slab_ranges(untouched)
        K.reduce_slabs(self.slabs, self.split_k, self.n_flat, self.n_flat, self.grad, scale=scale)
        return False

    def clip_and_adam(self, lr, max_norm=0.0, weight_decay=0.0):
        """nn.utils.clip_grad_norm_ + Adam.step over every parameter of the book."""
        self.step += 1
        K.sqnorm_partial(self.grad, self.n_flat, self.sq_partials)
        K.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.n_flat, lr=lr, step=self.step, weight_decay=weight_decay,
                    max_norm=max_norm, sqnorm_partials=self.sq_partials, grad_norm_out=self.grad_norm)


class Linear:
    """y[:, dst] = act(x[:, src] @ W^T + b).  ``in_width`` is the PHYSICAL input width (gaps included)."""

    def __init__(self, book, name, in_features, out_features, act=ACT_NONE, colmap=None, w_name=None, b_name=None, pitch_align=4):
        self.name, self.n, self.act = name, out_features, act
        self.w = book.add(w_name or f"{name}.weight", out_features, in_features, colmap, pitch_align=pitch_align)
        self.b = book.add(b_name or f"{name}.bias", 1, out_features)
        self.k_phys = (max(colmap) + 1) if colmap is not None else in_features
        self.k_logical = in_features


def init_linear_(book, lin, generator=None, zero_bias=True):
    """nn.Linear default init in the reference's order (weight, then the bias draw), bias zeroed as
    network_builder.py:273-277 / init_mlp do."""
    w = torch.empty(lin.n, lin.k_logical)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5), generator=generator)
    bound = 1 / math.sqrt(lin.k_logical)
    b = torch.empty(lin.n).uniform_(-bound, bound, generator=generator)
    book.set(lin.w.name, w)
    book.set(lin.b.name, torch.zeros(lin.n) if zero_bias else b)


class MlpGraph:
    """Activation / gradient buffers for a batch of m rows + forward / backward launch plans."""

    def __init__(self, book, m):
        self.book, self.m = book, m
        self.dev = book.device
        self.act_bufs, self.grad_bufs, self.pre_bufs = {}, {}, {}
        self.ops = []
        self._bias_scratch = None

    # ---- buffers ---------------------------------------------------------------------------------
    def buffer(self, name, width, tensor=None):
        if tensor is None:
            tensor = torch.zeros(self.m, r4(width), dtype=torch.float32, device=self.dev)
        self.act_bufs[name] = tensor
        return tensor

    def grad(self, name):
        if name not in self.grad_bufs:
            self.grad_bufs[name] = torch.zeros_like(self.act_bufs[name])
        return self.grad_bufs[name]

    def pre(self, name):
        if name not in self.pre_bufs:
            self.pre_bufs[name] = torch.zeros_like(self.act_bufs[name])
        return self.pre_bufs[name]

    # ---- graph construction ------------------------------------------------------------------------
    def linear(self, lin, src, dst, src_col=0, dst_col=0, grad_ranges=None, tag=None):
        """Register y = lin(x).  ``grad_ranges``: list of (c0, c1, act, aux_buffer_name, aux_col) column ranges of
        the SOURCE (relative to src_col) whose gradient is needed, with the activation that PRODUCED them
        (derivative fused into the dX GEMM).  None = no input gradient."""
        if src_col % 4 or dst_col % 4:
            raise ValueError("column offsets must be multiples of 4 floats (16-byte aligned segments)")
        self.ops.append({"lin": lin, "src": src, "dst": dst, "src_col": src_col, "dst_col": dst_col, "grad_ranges": grad_ranges or [],
                         "tag": tag})
        return lin

    def mlp(self, book, prefix, src, in_features, units, act, dst_names, src_col=0, colmap=None, first_grad_ranges=None, start_index=0,
            final_linear=None, final_dst=None, final_dst_col=0, tag=None):
        """nn.Sequential(Linear, act, Linear, act, ...) [+ an appended Linear without activation].  Layer i is
        named f"{prefix}.{start_index + 2 i}" like the reference's Sequential indices.  Returns the list of Linear objects."""
        lins, cur, cur_w, cur_col = [], src, in_features, src_col
        prev_act, prev_aux = None, None
        for i, u in enumerate(units):
            name = f"{prefix}.{start_index + 2 * i}"
            lin = Linear(book, name, cur_w, u, act, colmap if i == 0 else None)
            dst = dst_names[i]
            self.buffer(dst, u)
            if i == 0:
                gr = first_grad_ranges
            else:
                gr = [(0, cur_w, prev_act, prev_aux, 0)]
            self.linear(lin, cur, dst, src_col=cur_col if i == 0 else 0, grad_ranges=gr, tag=tag)
            lins.append(lin)
            prev_act, prev_aux = act, dst
            cur, cur_w, cur_col = dst, u, 0
        if final_linear is not None:
            name = f"{prefix}.{start_index + 2 * len(units)}"
            lin = Linear(book, name, cur_w, final_linear, ACT_NONE)
            if final_dst not in self.act_bufs:
                self.buffer(final_dst, final_linear)
            self.linear(lin, cur, final_dst, dst_col=final_dst_col, grad_ranges=[(0, cur_w, prev_act, prev_aux, 0)], tag=tag)
            lins.append(lin)
        return lins

    def untouched_ranges(self, tags):
        """Flat parameter ranges (merged, sorted) of the ops whose tag is NOT in ``tags``."""
        spans = []
        for op in self.ops:
            if op["tag"] in tags:
                continue
            lin = op["lin"]
            spans.append((lin.w.off, lin.w.off + lin.w.rows * lin.w.pitch))
            spans.append((lin.b.off, lin.b.off + lin.b.rows * lin.b.pitch))
        spans.sort()
        merged = []
        for lo, hi in spans:
            if merged and lo <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], hi))
            else:
                merged.append((lo, hi))
        return merged

    # ---- plans -----------------------------------------------------------------------------------------
    def forward_plan(self, tags=None, store_pre=True):
        """``store_pre=False``: inference only -- SiLU layers do not keep their pre-activation."""
        p = K.Plan()
        f = self.book.flat
        for op in self.ops:
            if tags is not None and op["tag"] not in tags:
                continue
            lin = op["lin"]
            x, y = self.act_bufs[op["src"]], self.act_bufs[op["dst"]]
            # a SiLU layer of a training pass keeps d silu / d z (ACT_SILU_D), not z: the input-gradient launch then multiplies (EPI_MUL_AUX)
            # instead of recomputing the sigmoid -- an exp and a division per element that the 256 x 256 x3 tile (one workgroup per CU) has
            # nothing to hide under (profiles/r05_gemm_shapes_cfg3.txt).  Same value, same bits: the derivative is the expression the
            # SILU_GRAD epilogue evaluates.  PULSE_SILU_DERIV=0: the pre-activation form.
            c2 = self.pre(op["dst"]) if (lin.act == ACT_SILU and store_pre) else None
            act = ACT_SILU_D if (c2 is not None and SILU_DERIV) else lin.act
            p.gemm(x, f, y, M=self.m, N=lin.n, K=lin.k_phys, lda=x.stride(0), ldb=lin.w.pitch, ldc=y.stride(0), bias=f,
                   activation=act, a_off=op["src_col"], b_off=lin.w.off, bias_off=lin.b.off, c_off=op["dst_col"],
                   C2=c2, ldc2=c2.stride(0) if c2 is not None else 0, c2_off=op["dst_col"], algo_k=lin.k_logical)
        return p

    def backward_plan(self, tags=None):
        """Consumes grad(dst) of every selected op (d loss / d pre-activation for ACT_NONE outputs, i.e. the caller
        seeds gradients of un-activated outputs; activated outputs receive theirs from their consumer's fused
        epilogue) and produces weight / bias gradients in the book's slabs."""
        p = K.Plan()
        book, f, m = self.book, self.book.flat, self.m
        S, P = book.split_k, book.n_flat
        if self._bias_scratch is None:
            width = max(r4(op["lin"].n) for op in self.ops)
            self._bias_scratch = torch.zeros(64, width + 8, dtype=torch.float32, device=self.dev)
            small = [op["lin"] for op in self.ops if op["lin"].n * op["lin"].w.pitch <= 128 * 1024]
            self._w_scratch = torch.zeros(32, r4(max([l.n * l.w.pitch + l.n for l in small] + [4])), dtype=torch.float32, device=self.dev)
        sc = self._bias_scratch
        for op in reversed(self.ops):
            if tags is not None and op["tag"] not in tags:
                continue
            lin = op["lin"]
            x = self.act_bufs[op["src"]]
            gz = self.grad(op["dst"])
            gzo, ldg = op["dst_col"], gz.stride(0)
            # weight gradient dW[n][k] = sum_m gz[m][n] x[m][k]; the bias gradient (column sums of gz) comes out of the same
            # launch as per-slab row sums of the A operand (pulse_gemm_desc.rowsum)
            tiles = ((lin.n + 127) // 128) * ((lin.k_phys + 127) // 128)
            wcount = lin.n * lin.w.pitch
            via_scratch = tiles * S < 128 and wcount + lin.n <= self._w_scratch.shape[1] and m >= 32 * 32 and lin.b.off == lin.w.off + wcount
            # how many gradient slabs this layer's launch writes (1: the scratch path sums into slab 0).  zero_slab_ranges only clears the
            # sub-networks a pass does NOT visit, so the slabs a visited layer leaves alone must never have been written by anyone: every
            # plan built over this book has to agree on the count (it depends on m) -- otherwise reduce_slabs would silently add another
            # graph's stale partial sums (round-3 advisor finding)
            # (the count is taken at the TRAINING minibatch's reduction length whatever m this graph is built for: graphs over one book must
            #  agree on it, and network_z / network_sept also build backward plans for the rollout's m -- round-5 advisor finding)
            sl = K.dw_split_x3(lin.n, lin.k_phys, 1, S, K=16384)
            book.note_slab_layout(lin.w.name, 1 if via_scratch else sl, m)
            book.note_slab_layout(lin.b.name, 1 if via_scratch else sl, m)      # (the bias gradient rides along as the launch's per-slab row sums)
            if via_scratch:
                ws_ = self._w_scratch
                p.gemm(gz, x, ws_, M=lin.n, N=lin.k_phys, K=m, lda=ldg, ldb=x.stride(0), ldc=lin.w.pitch, a_layout=GEMM_OUT_CONTIG,
                       b_layout=GEMM_OUT_CONTIG, a_off=gzo, b_off=op["src_col"], split_k=32, split_stride=ws_.stride(0), algo_n=lin.k_logical,
                       rowsum=ws_, rowsum_off=wcount)
                p.call("pulse_reduce_slabs", ws_.data_ptr(), 32, ws_.stride(0), wcount + lin.n, book.slabs.data_ptr() + 4 * lin.w.off, 1.0)
            else:
                p.gemm(gz, x, book.slabs, M=lin.n, N=lin.k_phys, K=m, lda=ldg, ldb=x.stride(0), ldc=lin.w.pitch, a_layout=GEMM_OUT_CONTIG,
                       b_layout=GEMM_OUT_CONTIG, a_off=gzo, b_off=op["src_col"], c_off=lin.w.off, split_k=sl, split_stride=P,
                       algo_n=lin.k_logical, rowsum=book.slabs, rowsum_off=lin.b.off)
            # input gradients for the requested column ranges, producer's activation derivative fused
            for (c0, c1, act, aux_name, aux_col) in op["grad_ranges"]:
                gx = self.grad(op["src"])
                epi, aux, aux_off, ldaux = EPI_BIAS_ACT, None, 0, 0
                if act == ACT_RELU:
                    epi, aux = EPI_RELU_GRAD, self.act_bufs[aux_name]
                elif act == ACT_SILU:
                    epi, aux = (EPI_MUL_AUX if SILU_DERIV else EPI_SILU_GRAD), self.pre(aux_name)
                if aux is not None:
                    aux_off, ldaux = aux_col + c0, aux.stride(0)
                p.gemm(gz, f, gx, M=m, N=c1 - c0, K=lin.n, lda=ldg, ldb=lin.w.pitch, ldc=gx.stride(0), b_layout=GEMM_OUT_CONTIG,
                       a_off=gzo, b_off=lin.w.off + c0, c_off=op["src_col"] + c0, epilogue=epi, aux=aux, ldaux=ldaux, aux_off=aux_off)
        return p
