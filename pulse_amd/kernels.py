"""Tensor-level wrappers for the learner-side C-ABI entries (GEMM, normaliser, PPO loss, Adam ...).

Same rules as ``ops.py``: GPU tensors only, enqueue on torch's current stream, no fallback.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, EPI_BIAS_ACT, GEMM_RED_CONTIG, GemmDesc, PpoLossArgs
from .ops import _dev, _ptr, _stream

ACTIVATIONS = {"None": ACT_NONE, "none": ACT_NONE, None: ACT_NONE, "relu": ACT_RELU, "silu": ACT_SILU}


def _p(t):
    return t.data_ptr() if t is not None else None


def _chk(t, name, dtype=torch.float32, contiguous=True):
    """Raw-pointer arguments: wrong dtype / device / layout would be silently re-interpreted by the kernel (e.g. uint8 dones read as
    int64), so they are rejected here.  None passes through (optional arguments)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise TypeError(f"{name}: expected a CUDA tensor, got {type(t).__name__} on {getattr(t, 'device', None)}")
    if t.dtype != dtype and not (dtype == torch.uint8 and t.dtype == torch.bool):
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and t.numel() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost stride must be 1, got strides {tuple(t.stride())}")
    return t.data_ptr()


class GemmProfiler:
    """Brackets every pulse_gemm_f32 launch with a HIP event pair on the launch stream (torch's
    current stream IS the stream the kernels are enqueued on) and tallies algorithmic FLOPs.
    Used by bench.py for the live roofline figure; off by default."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (start_event, end_event, flops, tag)
        self._pool = []

    def start(self):
        self.enabled, self.records = True, []

    def stop(self):
        self.enabled = False

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def summary(self):
        """-> dict tag -> (launches, seconds, flops); call after a synchronize."""
        out = {}
        for s, e, fl, tag in self.records:
            n, t, f = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, t + s.elapsed_time(e) * 1e-3, f + fl)
            self._pool += [s, e]
        self.records = []
        return out


PROFILER = GemmProfiler()

# How an fp32 GEMM is computed (inputs / outputs / storage are fp32 either way):
#   "x3"      three-way bf16 operand split, six bf16 MFMAs per k step, fp32 accumulation (PULSE_GEMM_COMPUTE_F32X3): fp32-grade
#             results on the bf16 matrix pipe, the fast path on gfx950
#   "mfma32"  v_mfma_f32_32x32x2_f32 (PULSE_GEMM_COMPUTE_F32)
F32_MODE = os.environ.get("PULSE_GEMM_F32", "x3")
if F32_MODE not in ("x3", "mfma32"):
    raise ValueError(f"PULSE_GEMM_F32 must be 'x3' or 'mfma32', not {F32_MODE!r}")


def make_gemm_desc(A, B, C, *, M, N, K, lda, ldb, ldc, a_layout=GEMM_RED_CONTIG, b_layout=GEMM_RED_CONTIG, bias=None,
                   activation=ACT_NONE, epilogue=EPI_BIAS_ACT, aux=None, ldaux=0, C2=None, ldc2=0, batch=1, stride_a=0,
                   stride_b=0, stride_c=0, stride_c2=0, stride_bias=0, stride_aux=0, split_k=1, split_stride=0,
                   a_off=0, b_off=0, c_off=0, bias_off=0, aux_off=0, c2_off=0, algo_k=None, algo_n=None, rowsum=None, rowsum_off=0,
                   stride_rowsum=0, compute_bf16=False, f32_mode=None, relu_mask=None, ld_mask=0, stride_mask=0, mask_off=0):
    """Build (descriptor, algorithmic FLOPs, variant tag) once; launch many times with launch_gemm.
    ``relu_mask``: int32 tensor (alloc_relu_mask) -- a relu forward records the sign bits there, a relu-grad launch with aux=None reads them
    (pulse_hip.h: pulse_gemm_desc.relu_mask); ld_mask / stride_mask / mask_off in 32-bit words."""
    d = GemmDesc()
    d.A = A.data_ptr() + 4 * a_off
    d.B = B.data_ptr() + 4 * b_off
    d.C = C.data_ptr() + 4 * c_off
    d.C2 = (C2.data_ptr() + 4 * c2_off) if C2 is not None else None
    d.bias = (bias.data_ptr() + 4 * bias_off) if bias is not None else None
    d.aux = (aux.data_ptr() + 4 * aux_off) if aux is not None else None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldc2, d.ldaux = lda, ldb, ldc, ldc2, ldaux
    d.a_layout, d.b_layout, d.batch = a_layout, b_layout, batch
    d.stride_a, d.stride_b, d.stride_c, d.stride_c2 = stride_a, stride_b, stride_c, stride_c2
    d.stride_bias, d.stride_aux = stride_bias, stride_aux
    d.split_k, d.split_stride, d.activation, d.epilogue = split_k, split_stride, activation, epilogue
    d.rowsum = (rowsum.data_ptr() + 4 * rowsum_off) if rowsum is not None else None
    d.stride_rowsum = stride_rowsum
    if relu_mask is not None:
        if relu_mask.dtype != torch.int32:
            raise TypeError("relu_mask must be an int32 tensor (alloc_relu_mask)")
        d.relu_mask, d.ld_mask, d.stride_mask = relu_mask.data_ptr() + 4 * mask_off, int(ld_mask), int(stride_mask)
    # bf16 MFMA with fp32 storage (bf16 autocast over fp32 master weights); split-K slabs are partial sums and stay unrounded
    x3 = (not compute_bf16) and (f32_mode or F32_MODE) == "x3"
    d.compute_type = _lib.GEMM_COMPUTE_BF16 if compute_bf16 else (_lib.GEMM_COMPUTE_F32X3 if x3 else _lib.GEMM_COMPUTE_F32)
    d.round_output_bf16 = 1 if (compute_bf16 and split_k == 1) else 0
    tag = ("fwd" if b_layout == GEMM_RED_CONTIG else "dx") if a_layout == GEMM_RED_CONTIG else "dw"
    if compute_bf16:
        tag = "bf16_" + tag
    elif x3:
        tag = "x3_" + tag
    flops = 2.0 * M * (algo_n if algo_n else N) * (algo_k if algo_k else K) * batch
    return d, flops, tag


def alloc_relu_mask(rows, cols, device):
    """The ReLU bit mask of a (rows, cols) activation matrix (pulse_gemm_desc.relu_mask): (roundup64(rows) / 8, roundup4(cols) / 4) int32,
    ld_mask = its row pitch in words.  1 bit per activation: 1/32 of the fp32 matrix a relu-grad epilogue would otherwise re-read."""
    return torch.zeros((rows + 63) // 64 * 8, (cols + 3) // 4, dtype=torch.int32, device=device)


def alloc_relu_mask8(rows, cols, device):
    """The ReLU bit mask of a (rows, cols) activation matrix in the planar kernels' layout (pulse_gemm_x3p_desc.relu_mask8): one byte per row
    and 8 columns."""
    return torch.zeros(rows, (cols + 7) // 8, dtype=torch.uint8, device=device)


def dw_split(tiles, max_split, fill=512):
    """Batch-split of a weight-gradient GEMM: the smallest power-of-two fraction of ``max_split`` (the slab count of the gradient
    buffer) that still gives ``fill`` workgroups = two per CU.  Fewer, longer reductions amortise each workgroup's prologue /
    epilogue (measured on the layer-1 dW of cfg2: 4 slabs 386 us, 8 slabs 404 us, 16 slabs 436 us); the slabs a layer does not
    write stay zero (allocated zeroed, never touched), so the ordered slab reduce is unchanged."""
    s = max_split
    while s > 1 and s % 2 == 0 and tiles * (s // 2) >= fill:
        s //= 2
    return s


def x3_tile_costs(M, N, z, k=16384):
    """(narrow, wide) cost of an x3 launch with z = batch x split_k grid slices, in the launcher's units (gemm_f32.hip: x3_wide_tile): one
    128 x 128 tile alone on a CU = 1.5 per unit of reduction length, two sharing a CU 2; a 256 x 256 workgroup has a CU to itself at about
    3.3 (more on short reductions, whose prologue / epilogue nothing hides)."""
    nt = ((M + 127) // 128) * ((N + 127) // 128) * z
    wt = ((M + 255) // 256) * ((N + 255) // 256) * z
    rem = nt % 512
    narrow = 2.0 * (nt // 512) + (0.0 if rem == 0 else 1.5 if rem <= 256 else 2.0)       # a lone workgroup is not twice as fast as a paired one
    wide_round = 2.0 * (0.1526 * k + 8.0) / (0.0924 * k + 5.0)          # k = the reduction length one workgroup walks (plain epilogue)
    wide = wide_round * ((wt + 255) // 256) if (M > 128 and N > 128) else float("inf")
    return narrow, wide


def dw_split_x3(M, N, batch, max_split, K=16384):
    """Slab count of an fp32 (x3) weight-gradient launch now that the launcher has two tilings: the power-of-two fraction of ``max_split`` with
    the lowest modelled time = min(narrow, wide cost) / split (the reduction length per workgroup is K / split) plus a charge per slab for its
    write and its share of the reduce (calibrated on the 2048 x 960 output: 16 us per 4 slabs against 281 us of GEMM), fewer slabs on ties.  cfg2 layer 1 (2048 x 960): 4 slabs on the narrow tiling -> 8 on the wide one
    (327 -> 296 us, profiles/r05_gemm_x3_wide_ab.txt); the [512, 1024] pair keeps 8 narrow slabs."""
    # the tiling the LAUNCHER will use (environment switch read once by the library + this thread's gemm option 4): planner and launcher
    # cannot disagree (round-5 advisor finding)
    if F32_MODE != "x3" or _lib.load().pulse_gemm_x3_mode() == 1:
        return dw_split(((M + 127) // 128) * ((N + 127) // 128) * batch, max_split)
    best, best_t = max_split, None
    s = max_split
    while s >= 1:
        t = min(x3_tile_costs(M, N, batch * s, max(16, K // s))) / s + 0.006 * s * (M * N * batch) / (2048.0 * 960.0)      # + the slab's write and its share of the reduce
        if best_t is None or t < best_t - 1e-9 or abs(t - best_t) <= 1e-9:
            best, best_t = s, t
        if s % 2:
            break
        s //= 2
    return best


def dw_split_b16(M, N, batch, max_split):
    """Slab count of a bf16-storage weight-gradient launch (pulse_gemm_x3p, planes = 1, both operands [red][out]).  The 256 x 256 tile moves
    2/3 of the bytes per MFMA of the 256 x 128 one (profiles/r04_ab_runs.txt: 2048 x 934 over 16384 rows: 124 us on 64 narrow tiles x 4
    slabs, 87 us on 32 wide tiles x 8 slabs), so when the wide tiling fills the chip within ``max_split`` slabs its count is returned -- the
    launcher's own rule (a wide workgroup must not cost a round) then picks that tile; otherwise the narrow tiling's count."""
    tm = (M + 255) // 256
    tw = tm * ((N + 255) // 256) * batch
    sw = dw_split(tw, max_split, fill=256)
    if N > 128 and tw * sw >= 192:
        return sw
    return dw_split(tm * ((N + 127) // 128) * batch, max_split, fill=256)


def launch_gemm(d, flops=0.0, tag="fwd", stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    if PROFILER.enabled:
        ev0, ev1 = PROFILER._event(), PROFILER._event()
        ev0.record()
        _lib.check(lib.pulse_gemm_f32(ctypes.byref(d), st), "pulse_gemm_f32")
        ev1.record()
        if tag.startswith("x3_") and lib.pulse_gemm_last_tile() == 256:
            tag = "x3w_" + tag[3:]            # served by the 256 x 256 tile (gemm_x3w_kernel): its own line in the bench's roofline
        PROFILER.records.append((ev0, ev1, flops, tag))
        return
    _lib.check(lib.pulse_gemm_f32(ctypes.byref(d), st), "pulse_gemm_f32")


def gemm(A, B, C, **kw):
    """C[m][n] = epilogue(sum_k A(m,k) B(n,k)).  A/B/C/... are tensors used only as base pointers
    (+ *_off floats); all geometry is explicit (pitches in floats)."""
    launch_gemm(*make_gemm_desc(A, B, C, **kw))


class Plan:
    """A pre-built launch sequence: GEMM descriptors and bound C-ABI calls are created once per
    workspace, so replaying a forward / backward pass costs one ctypes call per kernel."""

    def __init__(self, bf16=False):
        self.ops = []
        self.split = 0
        self.bf16 = bool(bf16)          # every GEMM of this plan runs on the bf16 MFMA (mixed_precision training passes)

    def gemm(self, A, B, C, **kw):
        kw.setdefault("compute_bf16", self.bf16)
        self.ops.append((0,) + make_gemm_desc(A, B, C, **kw))

    def call(self, name, *args):
        self.ops.append((1, getattr(_lib.load(), name), args, name))

    def call_partial_reduce(self, src, rows, count, slabs, dst_off):
        """Ordered sum of ``rows`` partial rows of ``src`` (row stride src.stride(0), ``count`` floats each) into slab 0 of ``slabs`` at element
        ``dst_off`` (pulse_reduce_slabs).  Registered separately: run(skip_partial_reduces=True) leaves these launches out when the caller's
        ReduceGrads reads the partials itself (``partial_reduces`` lists what it has to read)."""
        self.ops.append((3, _lib.load().pulse_reduce_slabs, (src.data_ptr(), int(rows), int(src.stride(0)), int(count), slabs.data_ptr() + 4 * int(dst_off), 1.0),
                         "pulse_reduce_slabs"))
        if not hasattr(self, "partial_reduces"):
            self.partial_reduces = []
        self.partial_reduces.append((int(dst_off), int(count), int(rows), src))

    def gemm_x3p(self, A, B, **kw):
        self.ops.append((2,) + make_gemm_x3p_desc(A, B, **kw))

    def gemm_b16(self, A, B, **kw):
        """bf16-storage GEMM (pulse_gemm_x3p, planes = 1): A / B / Cp / aux are int16 (bf16 bit pattern) tensors."""
        self.ops.append((2,) + make_gemm_x3p_desc(A, B, planes=1, **kw))

    def refresh_b16(self, flat, flat16, count):
        """flat16[i] = bf16(flat[i]) for the whole flat parameter buffer: the bf16 weight image the plan's GEMMs read is rebuilt inside the
        plan, so no writer of the parameters (optimiser step, checkpoint load, broadcast, a test poking the buffer) can leave it stale."""
        n8 = (count + 7) // 8 * 8
        if flat16.dtype != torch.int16 or flat16.numel() < n8 or flat.numel() < count:
            raise ValueError("refresh_b16: flat16 must be an int16 buffer of roundup8(count) elements")
        self.call("pulse_split_planes", flat.data_ptr(), count, 1, count, flat16.data_ptr(), 0, n8, 0, None)

    def weights_b16(self, flat, flat16, count, transposes=()):
        """refresh_b16 + up to four transpose_b16 in ONE launch (pulse_weights_to_b16); ``transposes``: dicts with the keyword arguments of
        transpose_b16 plus ``x`` / ``out``."""
        n8 = (count + 7) // 8 * 8
        if flat16.dtype != torch.int16 or flat16.numel() < n8 or flat.numel() < count:
            raise ValueError("weights_b16: flat16 must be an int16 buffer of roundup8(count) elements")
        if len(transposes) > 4:
            raise ValueError("weights_b16: at most four transposes per launch")
        arr = (_lib.B16Transpose * max(1, len(transposes)))()
        for d, kw in zip(arr, transposes):
            if kw["out"].dtype != torch.int16:
                raise TypeError("weights_b16: transposed images must be int16")
            d.in_, d.ld_in, d.rows, d.cols = kw["x"].data_ptr() + 4 * kw.get("x_off", 0), kw["ld_in"], kw["rows"], kw["cols"]
            d.out, d.ld_out = kw["out"].data_ptr() + 2 * kw.get("out_off", 0), kw["ld_out"]
            d.batch, d.stride_in, d.stride_out = kw.get("batch", 1), kw.get("stride_in", 0), kw.get("stride_out", 0)
        self._keep = getattr(self, "_keep", []) + [arr]
        self.call("pulse_weights_to_b16", flat.data_ptr(), count, flat16.data_ptr(), len(transposes), arr)

    def transpose_b16(self, x, out, **kw):
        x_off, out_off = kw.pop("x_off", 0), kw.pop("out_off", 0)
        if out.dtype != torch.int16:
            raise TypeError("transpose_b16: out must be int16")
        self.call("pulse_transpose_to_b16", x.data_ptr() + 4 * x_off, kw["ld_in"], kw["rows"], kw["cols"], out.data_ptr() + 2 * out_off, kw["ld_out"],
                  kw.get("batch", 1), kw.get("stride_in", 0), kw.get("stride_out", 0))

    def colsum_b16(self, x, m, n, ld, slabs, num_slabs, slab_stride, out_off, x_off=0):
        """Bias gradient of a bf16 gradient matrix (m, n): slab s of ``slabs`` receives, at [out_off, out_off + n), the column sums over the
        s-th of ``num_slabs`` row ranges -- summed with the weight gradients by the one slab reduce, like the fp32 kernels' ``rowsum``."""
        if x.dtype != torch.int16:
            raise TypeError("colsum_b16: x must be an int16 (bf16 bit pattern) tensor")
        self.call("pulse_colsum_partial_b16", x.data_ptr() + 2 * x_off, m, n, ld, num_slabs, slabs.data_ptr() + 4 * out_off, slab_stride)

    def colsum_weighted_b16(self, x, w16, w_stride, m, n, ld, slabs, num_slabs, slab_stride, out_off, w_off=0):
        """slab s [out_off, out_off + n) = sum over the s-th row range of w16[m * w_stride] * x[m][:]: the weight gradient of a one-output Linear."""
        if x.dtype != torch.int16 or w16.dtype != torch.int16:
            raise TypeError("colsum_weighted_b16: x / w16 must be int16 (bf16 bit pattern) tensors")
        self.call("pulse_colsum_weighted_b16", x.data_ptr(), m, n, ld, w16.data_ptr() + 2 * w_off, w_stride, num_slabs, slabs.data_ptr() + 4 * out_off,
                  slab_stride)

    def run(self, start=0, stop=None, skip_partial_reduces=False):
        """``skip_partial_reduces``: leave out the small ordered reduces registered with call_partial_reduce -- the caller's fused gradient reduce
        (ReduceGrads regions with their own source) sums those partials itself.
        (Round 5 measured replaying plan segments as captured HIP graphs: flat inside the epoch -- cfg2 68.0 / 68.6 ms eager vs 68.4 / 69.2,
        profiles/r05_ab_runs.txt; the update is one dependent chain the host enqueues far ahead of the device.  The path was removed in round 6.)"""
        st = _stream()
        for op in self.ops[start:stop]:
            if op[0] == 0:
                launch_gemm(op[1], op[2], op[3], st)
            elif op[0] == 2:
                launch_gemm_x3p(op[1], op[2], op[3], st)
            elif op[0] == 3:
                if not skip_partial_reduces:
                    _lib.check(op[1](*op[2], st), op[3])
            else:
                _lib.check(op[1](*op[2], st), op[3])


# --------------------------------------------------------------------------- #
# PULSE VAE head algebra (include/pulse_hip.h section 4c)
# --------------------------------------------------------------------------- #
def vae_embed(heads, x, ain, *, rows, embedding_size, self_obs_size, z_col, eps=None, cin=None, clamp=True, clamp_max=2.0):
    a = _lib.VaeEmbedArgs()
    a.heads, a.heads_stride = _chk(heads, "heads"), heads.stride(0)
    a.eps, a.eps_stride = _chk(eps, "eps"), (eps.stride(0) if eps is not None else 0)
    a.x, a.x_stride = _chk(x, "x"), x.stride(0)
    a.ain, a.ain_stride = _chk(ain, "ain"), ain.stride(0)
    a.cin, a.cin_stride = _chk(cin, "cin"), (cin.stride(0) if cin is not None else 0)
    a.rows, a.embedding_size, a.self_obs_size, a.z_col = rows, embedding_size, self_obs_size, z_col
    a.clamp_logvar, a.clamp_max = int(bool(clamp)), float(clamp_max)
    _lib.check(_lib.load().pulse_vae_embed(ctypes.byref(a), _stream()), "pulse_vae_embed")


def vae_kin_loss(pred, gt, zheads, pheads, progress, dmu, partials, *, rows, num_actions, embedding_size, horizon, clamp=True, clamp_max=2.0,
                 use_ar1=True, use_regu=False):
    a = _lib.VaeKinArgs()
    a.pred, a.pred_stride, a.gt, a.gt_stride = _chk(pred, "pred"), pred.stride(0), _chk(gt, "gt"), gt.stride(0)
    a.zheads, a.zheads_stride, a.pheads, a.pheads_stride = _chk(zheads, "zheads"), zheads.stride(0), _chk(pheads, "pheads"), pheads.stride(0)
    a.progress = _chk(progress, "progress", torch.int64)
    a.rows, a.num_actions, a.embedding_size, a.horizon = rows, num_actions, embedding_size, horizon
    a.clamp_logvar, a.clamp_max, a.use_ar1, a.use_regu = int(bool(clamp)), float(clamp_max), int(bool(use_ar1)), int(bool(use_regu))
    a.dmu, a.dmu_stride = _chk(dmu, "dmu"), dmu.stride(0)
    a.partials, a.num_blocks = _chk(partials, "partials"), partials.shape[0]
    _lib.check(_lib.load().pulse_vae_kin_loss(ctypes.byref(a), _stream()), "pulse_vae_kin_loss")


def vae_head_backward(zheads, dzheads, *, rows, embedding_size, horizon=1, pheads=None, dpheads=None, eps=None, dz=None, progress=None, clamp=True,
                      clamp_max=2.0, c_kl=0.0, c_ar1=0.0, c_regu=0.0):
    a = _lib.VaeHeadBwdArgs()
    a.zheads, a.zheads_stride = _chk(zheads, "zheads"), zheads.stride(0)
    a.pheads, a.pheads_stride = _chk(pheads, "pheads"), (pheads.stride(0) if pheads is not None else 0)
    a.eps, a.eps_stride = _chk(eps, "eps"), (eps.stride(0) if eps is not None else 0)
    a.dz, a.dz_stride = _chk(dz, "dz"), (dz.stride(0) if dz is not None else 0)
    a.progress = _chk(progress, "progress", torch.int64)
    a.rows, a.embedding_size, a.horizon = rows, embedding_size, horizon
    a.clamp_logvar, a.clamp_max = int(bool(clamp)), float(clamp_max)
    a.c_kl, a.c_ar1, a.c_regu = float(c_kl), float(c_ar1), float(c_regu)
    a.dzheads, a.dzheads_stride = _chk(dzheads, "dzheads"), dzheads.stride(0)
    a.dpheads, a.dpheads_stride = _chk(dpheads, "dpheads"), (dpheads.stride(0) if dpheads is not None else 0)
    _lib.check(_lib.load().pulse_vae_head_backward(ctypes.byref(a), _stream()), "pulse_vae_head_backward")


# --------------------------------------------------------------------------- #
# planar ("x3p") fp32-grade GEMM: operands kept pre-split in HBM as three bf16 planes (include/pulse_hip.h section 4b)
# --------------------------------------------------------------------------- #
def planes_pitch(cols):
    """Element pitch of a planes row: the k extent is zero-padded to a multiple of 32 inside the pitch."""
    return (cols + 31) // 32 * 32


def alloc_planes(rows, cols, device):
    """(3, rows, pitch) int16 (bf16 bit patterns), zero-initialised: the pad columns must read as zero."""
    return torch.zeros(3, rows, planes_pitch(cols), dtype=torch.int16, device=device)


def split_planes(x, out=None, *, rows=None, cols=None, transpose=False, row_idx=None, x_off=0, ld_in=None, out_off=0):
    """fp32 matrix -> its three bf16 planes (x == p0 + p1 + p2 exactly).  ``x`` is a 2-D tensor (or a base tensor with x_off / ld_in /
    rows / cols given explicitly); transpose=True writes the planes of x.T."""
    _chk(x, "x")
    if ld_in is None:
        if x.dim() != 2 or x.stride(1) != 1:
            raise ValueError("split_planes: 2-D tensor with contiguous rows expected (or explicit geometry)")
        ld_in = x.stride(0)
        r_in, c_in = x.shape
        rows_out, cols_out = (c_in, r_in) if transpose else (r_in, c_in)
        if row_idx is not None:
            rows_out = row_idx.numel()
    else:
        rows_out, cols_out = rows, cols
    if out is None:
        out = alloc_planes(rows_out, cols_out, x.device)
    if out.dtype != torch.int16 or out.dim() != 3 or out.shape[0] != 3 or out.stride(2) != 1 or not out.is_cuda:
        raise TypeError("split_planes: out must be a (3, rows, pitch) int16 CUDA tensor")
    _chk(row_idx, "row_idx", torch.int64)
    _lib.check(_lib.load().pulse_split_planes(x.data_ptr() + 4 * x_off, ld_in, rows_out, cols_out, out.data_ptr() + 2 * out_off, out.stride(0),
                                              out.stride(1), 1 if transpose else 0, _p(row_idx), _stream()), "pulse_split_planes")
    return out


def join_planes(p):
    """The fp32 matrix a planes tensor represents (test helper): p0 + p1 + p2 evaluated exactly."""
    f = (p.to(torch.int32) << 16).view(torch.float32)
    return (f[0].double() + f[1].double() + f[2].double()).float()


def alloc_b16(rows, cols, device):
    """(rows, pitch) int16 (bf16 bit patterns), zero-initialised: a single-plane operand / output of gemm_x3p(planes=1)."""
    return torch.zeros(rows, planes_pitch(cols), dtype=torch.int16, device=device)


def to_b16(x, out=None, *, rows=None, cols=None, transpose=False, row_idx=None, x_off=0, ld_in=None, out_off=0):
    """fp32 matrix -> the same matrix rounded to bf16 (round to nearest even), pad columns zero-filled.  Geometry arguments as split_planes."""
    _chk(x, "x")
    if ld_in is None:
        if x.dim() != 2 or x.stride(1) != 1:
            raise ValueError("to_b16: 2-D tensor with contiguous rows expected (or explicit geometry)")
        ld_in = x.stride(0)
        r_in, c_in = x.shape
        rows_out, cols_out = (c_in, r_in) if transpose else (r_in, c_in)
        if row_idx is not None:
            rows_out = row_idx.numel()
    else:
        rows_out, cols_out = rows, cols
    if out is None:
        out = alloc_b16(rows_out, cols_out, x.device)
    if out.dtype != torch.int16 or out.dim() != 2 or out.stride(1) != 1 or not out.is_cuda:
        raise TypeError("to_b16: out must be a (rows, pitch) int16 CUDA tensor")
    _chk(row_idx, "row_idx", torch.int64)
    _lib.check(_lib.load().pulse_split_planes(x.data_ptr() + 4 * x_off, ld_in, rows_out, cols_out, out.data_ptr() + 2 * out_off, 0,
                                              out.stride(0), 1 if transpose else 0, _p(row_idx), _stream()), "pulse_split_planes")
    return out


def from_b16(p):
    """The fp32 values of a bf16 bit-pattern tensor."""
    return (p.to(torch.int32) << 16).view(torch.float32)


def make_gemm_x3p_desc(A, B, *, M, N, K, C=None, Cp=None, bias=None, activation=ACT_NONE, epilogue=EPI_BIAS_ACT, aux=None, ldaux=0, C2=None,
                       ldc2=0, ldc=0, batch=1, stride_a=0, stride_b=0, stride_c=0, stride_cp=0, stride_c2=0, stride_bias=0, stride_aux=0,
                       a_off=0, b_off=0, c_off=0, cp_off=0, bias_off=0, aux_off=0, c2_off=0, algo_k=None, algo_n=None, planes=3,
                       a_layout=GEMM_RED_CONTIG, b_layout=GEMM_RED_CONTIG, split_k=1, split_stride=0, lda=None, ldb=None, ldcp=None,
                       out_colsum=None, out_colsum_off=0, stride_out_colsum=0, ld_out_colsum=None, relu_mask8=None, ld_mask8=None, stride_mask8=0,
                       mask8_off=0):
    """planes=3: A / B / Cp are planes tensors (3, rows, pitch) int16.  planes=1 (bf16 operands): (rows, pitch) int16 matrices (or flat
    int16 buffers with lda / ldb / ldcp given), aux may be one too (ReLU-mask / SiLU-derivative epilogues reading bf16 activations).
    *_off are element offsets.  Returns (descriptor, algorithmic FLOPs, tag) like make_gemm_desc."""
    nd = 3 if planes == 3 else 2
    for t, nm, ld in ((A, "A", lda), (B, "B", ldb), (Cp, "Cp", ldcp)):
        if t is None:
            continue
        flat_ok = planes == 1 and ld is not None and t.dim() == 1
        if t.dtype != torch.int16 or not t.is_cuda or not (flat_ok or (t.dim() == nd and t.stride(nd - 1) == 1 and (planes == 1 or t.shape[0] == 3))):
            raise TypeError(f"gemm_x3p: {nm} must be a {'(3, rows, pitch)' if planes == 3 else '(rows, pitch)'} int16 CUDA tensor")
    d = _lib.GemmX3pDesc()
    d.planes = planes
    d.A, d.a_plane_stride, d.lda = A.data_ptr() + 2 * a_off, (A.stride(0) if planes == 3 else 0), (lda if lda is not None else A.stride(nd - 2))
    d.B, d.b_plane_stride, d.ldb = B.data_ptr() + 2 * b_off, (B.stride(0) if planes == 3 else 0), (ldb if ldb is not None else B.stride(nd - 2))
    d.a_layout, d.b_layout = a_layout, b_layout
    if C is not None:
        _chk(C, "C")
        d.C, d.ldc = C.data_ptr() + 4 * c_off, ldc
    if Cp is not None:
        d.Cp, d.c_plane_stride, d.ldcp = Cp.data_ptr() + 2 * cp_off, (Cp.stride(0) if planes == 3 else 0), (ldcp if ldcp is not None else Cp.stride(nd - 2))
    d.C2 = (C2.data_ptr() + 4 * c2_off) if C2 is not None else None
    d.ldc2 = ldc2
    d.bias = (bias.data_ptr() + 4 * bias_off) if bias is not None else None
    if aux is not None and aux.dtype == torch.int16:
        d.aux, d.aux_is_bf16 = aux.data_ptr() + 2 * aux_off, 1
    else:
        d.aux = (aux.data_ptr() + 4 * aux_off) if aux is not None else None
    d.ldaux = ldaux
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.stride_a, d.stride_b, d.stride_c, d.stride_cp, d.stride_c2 = stride_a, stride_b, stride_c, stride_cp, stride_c2
    d.stride_bias, d.stride_aux = stride_bias, stride_aux
    d.split_k, d.split_stride, d.activation, d.epilogue = split_k, split_stride, activation, epilogue
    if out_colsum is not None:             # (row_tiles(M, N, batch), >= covering columns) fp32: per-row-tile column sums of the stored output
        _chk(out_colsum, "out_colsum")
        ld = out_colsum.stride(0) if ld_out_colsum is None else ld_out_colsum
        need = gemm_x3p_row_tiles(M, N, batch)
        if out_colsum.dim() != 2 or out_colsum.shape[0] < need:
            raise ValueError(f"gemm_x3p: out_colsum needs {need} rows (one per row tile)")
        d.out_colsum, d.stride_out_colsum, d.ld_out_colsum = out_colsum.data_ptr() + 4 * out_colsum_off, stride_out_colsum, ld
    if relu_mask8 is not None:             # uint8 (rows, roundup8(cols) / 8): alloc_relu_mask8; offsets / strides in bytes
        if relu_mask8.dtype != torch.uint8:
            raise TypeError("relu_mask8 must be a uint8 tensor (alloc_relu_mask8)")
        d.relu_mask8 = relu_mask8.data_ptr() + int(mask8_off)
        d.ld_mask8, d.stride_mask8 = int(relu_mask8.stride(0) if ld_mask8 is None else ld_mask8), int(stride_mask8)
    flops = 2.0 * M * (algo_n if algo_n else N) * (algo_k if algo_k else K) * batch
    kc_a, kc_b = a_layout == GEMM_RED_CONTIG, b_layout == GEMM_RED_CONTIG
    tag = ("x3p_" if planes == 3 else "b16_") + ("fwd" if kc_a and kc_b else "dx" if kc_a else "dw")
    return d, flops, tag


def gemm_x3p_row_tiles(M, N, batch=1):
    """Row tiles (256 or 128 rows) a pulse_gemm_x3p launch of this shape uses = rows of its out_colsum partials."""
    return int(_lib.load().pulse_gemm_x3p_row_tiles(int(M), int(N), int(batch)))


def gemm_set_option(key, value):
    """Diagnostics knob of the calling host thread (pulse_hip.h, section 4): tests and tools only -- the product path never sets one."""
    _lib.check(_lib.load().pulse_gemm_set_option(int(key), int(value)), "pulse_gemm_set_option")


def launch_gemm_x3p(d, flops=0.0, tag="x3p_fwd", stream=None):
    lib = _lib.load()
    st = _stream() if stream is None else stream
    if PROFILER.enabled:
        ev0, ev1 = PROFILER._event(), PROFILER._event()
        ev0.record()
        _lib.check(lib.pulse_gemm_x3p(ctypes.byref(d), st), "pulse_gemm_x3p")
        ev1.record()
        PROFILER.records.append((ev0, ev1, flops, tag))
        return
    _lib.check(lib.pulse_gemm_x3p(ctypes.byref(d), st), "pulse_gemm_x3p")


def gemm_x3p(A, B, **kw):
    launch_gemm_x3p(*make_gemm_x3p_desc(A, B, **kw))


def linear_forward(x, w, bias=None, activation=ACT_NONE, out=None):
    """Convenience: y = act(x @ w.T + bias) for 2-D row-major tensors with 16-byte aligned rows."""
    x, w = _dev(x, "x"), _dev(w, "w")
    m, k = x.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    gemm(x, w, out, M=m, N=n, K=k, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=bias, activation=activation)
    return out


def reduce_slabs(slabs, num_slabs, slab_stride, count, out, scale=1.0, slabs_off=0, out_off=0):
    _lib.check(_lib.load().pulse_reduce_slabs(slabs.data_ptr() + 4 * slabs_off, num_slabs, slab_stride, count,
                                              out.data_ptr() + 4 * out_off, float(scale), _stream()), "pulse_reduce_slabs")


def colsum_partial(x, m, n, ld, num_chunks, partial, ld_partial, x_off=0, partial_off=0):
    _lib.check(_lib.load().pulse_colsum_partial(x.data_ptr() + 4 * x_off, m, n, ld, num_chunks,
                                                partial.data_ptr() + 4 * partial_off, ld_partial, _stream()), "pulse_colsum_partial")


def rms_copy_supported(x, cols, y, y_cols, raw_out):
    """Can ``rms_normalize(..., raw_out=raw_out)`` run (pulse_rms_normalize_copy's wide-row form)?  Decided by the caller once per buffer set."""
    ok = lambda t: t is not None and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    c4 = (cols + 3) // 4 * 4
    return bool(ok(x) and ok(y) and ok(raw_out) and 64 <= cols and y_cols <= 3072 and y_cols % 4 == 0 and y_cols >= cols and x.stride(0) >= c4 and
                y.stride(0) >= y_cols and raw_out.stride(0) >= c4 and raw_out.shape[1] >= c4)


def rms_normalize(x, mean, var, *, rows, cols, x_stride, y, y_stride, y_cols=None, row_idx=None, eps=1e-5, clip=5.0,
                  unnorm=False, moment_partials=None, num_blocks=None, planes=None, raw_out=None):
    """``planes``: a (3, rows, pitch) int16 planes tensor that also receives the exact three-way bf16 split of y (layer-1 operand of gemm_x3p).
    ``y`` of dtype int16: the output IS a bf16 matrix (layer-1 operand of the bf16-storage training path); wide rows only.
    ``raw_out``: a (rows, >= cols rounded up to 4) float32 view (any 16-byte aligned row pitch) that also receives the RAW rows read (the rollout's
    record of the observation, see rms_copy_supported); normalising direction, fp32 output, no planes."""
    if num_blocks is None:
        num_blocks = max(1, min(512, rows // 16)) if moment_partials is None else moment_partials.shape[0]
    if raw_out is not None:
        if unnorm or planes is not None or y.dtype != torch.float32:
            raise ValueError("rms_normalize: raw_out goes with the plain normalising direction only")
        if not raw_out.is_cuda or raw_out.dtype != torch.float32 or raw_out.dim() != 2 or raw_out.stride(1) != 1 or raw_out.shape[0] < rows:
            raise TypeError("rms_normalize: raw_out must be a (>= rows, cols) float32 CUDA view with unit inner stride")
        _lib.check(_lib.load().pulse_rms_normalize_copy(_p(x), x_stride, _p(row_idx), rows, cols, _p(mean), _p(var), eps, clip, _p(y), y_stride,
                                                        cols if y_cols is None else y_cols, _p(moment_partials), num_blocks, raw_out.data_ptr(),
                                                        raw_out.stride(0), _stream()), "pulse_rms_normalize_copy")
        return
    if y.dtype == torch.int16:
        if unnorm or planes is not None or not y.is_cuda or y.stride(-1) != 1:
            raise ValueError("rms_normalize: a bf16 output goes with the normalising direction, without planes")
        _lib.check(_lib.load().pulse_rms_normalize_b16(_p(x), x_stride, _p(row_idx), rows, cols, _p(mean), _p(var), eps, clip, y.data_ptr(), y_stride,
                                                       cols if y_cols is None else y_cols, _p(moment_partials), num_blocks, _stream()), "pulse_rms_normalize_b16")
        return
    if planes is not None:
        if unnorm:
            raise ValueError("rms_normalize: planes go with the normalising direction only")
        if planes.dtype != torch.int16 or planes.dim() != 3 or planes.shape[0] != 3 or planes.stride(2) != 1 or planes.shape[1] < rows or not planes.is_cuda:
            raise TypeError("rms_normalize: planes must be a (3, >= rows, pitch) int16 CUDA tensor")
        _lib.check(_lib.load().pulse_rms_normalize_planes(_p(x), x_stride, _p(row_idx), rows, cols, _p(mean), _p(var), eps, clip, _p(y), y_stride,
                                                          cols if y_cols is None else y_cols, _p(moment_partials), num_blocks, planes.data_ptr(),
                                                          planes.stride(0), planes.stride(1), _stream()), "pulse_rms_normalize_planes")
        return
    _lib.check(_lib.load().pulse_rms_normalize(_p(x), x_stride, _p(row_idx), rows, cols, _p(mean), _p(var), eps, clip,
                                               1 if unnorm else 0, _p(y), y_stride, cols if y_cols is None else y_cols,
                                               _p(moment_partials), num_blocks, _stream()), "pulse_rms_normalize")


def rms_update(mean, var, count, moment_partials, cols, count_old, batch_count):
    _lib.check(_lib.load().pulse_rms_update(_p(mean), _p(var), _p(count), _p(moment_partials), moment_partials.shape[0], cols,
                                            float(count_old), float(batch_count), _stream()), "pulse_rms_update")


def policy_sample(mu, mu_stride, logstd, noise, noise_stride, rows, num_actions, actions, actions_stride, neglogp,
                  neglogp_stride=1, sigmas=None, sigmas_stride=0, value_raw=None, value_stride=0, value_mean=None,
                  value_var=None, values=None, values_stride=0, mu_off=0, actions_off=0, sigmas_off=0, neglogp_off=0,
                  values_off=0, mus_out=None, mus_out_stride=0, mus_out_off=0):
    def po(t, off):
        return (t.data_ptr() + 4 * off) if t is not None else None
    for t, nm in ((mu, "mu"), (logstd, "logstd"), (noise, "noise"), (value_raw, "value_raw"), (actions, "actions"), (sigmas, "sigmas"),
                  (neglogp, "neglogp"), (values, "values"), (mus_out, "mus_out")):
        _chk(t, nm)
    _chk(value_mean, "value_mean", torch.float64), _chk(value_var, "value_var", torch.float64)
    _lib.check(_lib.load().pulse_policy_sample(po(mu, mu_off), mu_stride, _p(logstd), _p(noise), noise_stride, _p(value_raw),
                                               value_stride, _p(value_mean), _p(value_var), rows, num_actions,
                                               po(actions, actions_off), actions_stride, po(sigmas, sigmas_off), sigmas_stride,
                                               po(neglogp, neglogp_off), neglogp_stride, po(values, values_off), values_stride,
                                               po(mus_out, mus_out_off), mus_out_stride, _stream()), "pulse_policy_sample")


def ppo_loss(*, mu, mu_stride, value, value_stride, logstd, old_logstd, idx, actions, actions_stride, old_mu, old_mu_stride,
             old_neglogp, advantages, old_values, returns, rows, num_actions, e_clip, critic_coef, bounds_loss_coef, clip_value,
             dmu, dmu_stride, dvalue, dvalue_stride, partials, dmu16=None, dmu16_stride=0, dvalue16=None, dvalue16_stride=0, dmu16_off=0, dvalue16_off=0):
    a = PpoLossArgs()
    a.mu, a.mu_stride, a.value, a.value_stride, a.logstd = _p(mu), mu_stride, _p(value), value_stride, _p(logstd)
    a.idx, a.actions, a.actions_stride = _p(idx), _p(actions), actions_stride
    a.old_mu, a.old_mu_stride, a.old_logstd = _p(old_mu), old_mu_stride, _p(old_logstd)
    a.old_neglogp, a.advantages, a.old_values, a.returns = _p(old_neglogp), _p(advantages), _p(old_values), _p(returns)
    a.rows, a.num_actions = rows, num_actions
    a.e_clip, a.critic_coef = float(e_clip), float(critic_coef)
    a.has_bounds_loss = 0 if bounds_loss_coef is None else 1
    a.bounds_loss_coef = 0.0 if bounds_loss_coef is None else float(bounds_loss_coef)
    a.clip_value = 1 if clip_value else 0
    a.dmu, a.dmu_stride, a.dvalue, a.dvalue_stride = _p(dmu), dmu_stride, _p(dvalue), dvalue_stride
    a.partials, a.num_blocks = _p(partials), partials.shape[0]
    if dmu16 is not None:                  # bf16 copies of the head gradients (int16 bit patterns; *_off in bf16 elements)
        if dmu16.dtype != torch.int16 or dvalue16 is None or dvalue16.dtype != torch.int16:
            raise TypeError("ppo_loss: dmu16 / dvalue16 must be int16 (bf16 bit pattern) CUDA tensors")
        a.dmu16, a.dmu16_stride = dmu16.data_ptr() + 2 * dmu16_off, dmu16_stride
        a.dvalue16, a.dvalue16_stride = dvalue16.data_ptr() + 2 * dvalue16_off, dvalue16_stride
    _lib.check(_lib.load().pulse_ppo_loss(ctypes.byref(a), _stream()), "pulse_ppo_loss")


def advantage_normalize(returns, values, adv_out, partials):
    """adv = (returns - values - mean) / (std + 1e-8) over flat tensors (common_agent.py:589-599)."""
    n = returns.numel()
    lib = _lib.load()
    _lib.check(lib.pulse_advantage_moments(_p(returns), _p(values), n, _p(adv_out), _p(partials), partials.shape[0], _stream()),
               "pulse_advantage_moments")
    _lib.check(lib.pulse_advantage_normalize(_p(adv_out), n, _p(partials), partials.shape[0], _stream()), "pulse_advantage_normalize")
    return adv_out


def sqnorm_partial(x, count, partials):
    _chk(x, "x"), _chk(partials, "partials")
    if x.numel() < count or not x.is_contiguous():
        raise ValueError(f"sqnorm_partial: x must be a contiguous buffer of at least {count} floats")
    _lib.check(_lib.load().pulse_sqnorm_partial(_p(x), count, _p(partials), partials.numel(), _stream()), "pulse_sqnorm_partial")


def disc_head(logits, b, scale, dlogits, stats):
    """AMPAgent._disc_loss head on the (3b, 1) logit column (any row stride): fills ``dlogits`` (same shape) and ``stats`` (8 floats:
    pred loss, agent BCE, demo BCE, agent acc, demo acc, agent logit mean, demo logit mean, 0)."""
    _chk(logits, "logits"), _chk(dlogits, "dlogits"), _chk(stats, "stats")
    if logits.shape[0] != 3 * b or dlogits.shape[0] != 3 * b or stats.numel() < 8 or not stats.is_contiguous():
        raise ValueError("disc_head: logits / dlogits must have 3b rows and stats 8 contiguous floats")
    _lib.check(_lib.load().pulse_disc_head(_p(logits), logits.stride(0), b, float(scale), _p(dlogits), dlogits.stride(0), _p(stats), _stream()),
               "pulse_disc_head")


def disc_head_b16(logits, b, scale, dlogits16, stats, dlogits=None, bias_grad=None):
    """disc_head writing the logit gradients as bf16 (dlogits16: int16 (>= 3b, pitch) tensor, column 0) and optionally as fp32 too;
    ``bias_grad`` (a float32 view, element 0): receives their sum = the logit bias' gradient."""
    _chk(logits, "logits"), _chk(stats, "stats"), _chk(dlogits, "dlogits"), _chk(bias_grad, "bias_grad", contiguous=False)
    if dlogits16.dtype != torch.int16 or not dlogits16.is_cuda or dlogits16.shape[0] < 3 * b or logits.shape[0] != 3 * b or stats.numel() < 8:
        raise ValueError("disc_head_b16: logits must have 3b rows, dlogits16 be an int16 CUDA tensor of >= 3b rows, stats 8 floats")
    _lib.check(_lib.load().pulse_disc_head_b16(_p(logits), logits.stride(0), b, float(scale), _p(dlogits), dlogits.stride(0) if dlogits is not None else 0,
                                               dlogits16.data_ptr(), dlogits16.stride(0), _p(stats), _p(bias_grad), _stream()), "pulse_disc_head_b16")


def transpose_to_b16(x, out, *, rows, cols, ld_in, ld_out, batch=1, stride_in=0, stride_out=0, x_off=0, out_off=0):
    """out[z][c][r] = bf16(x[z][r][c]); x fp32 base tensor (+ x_off floats), out int16 base tensor (+ out_off elements)."""
    _chk(x, "x")
    if out.dtype != torch.int16 or not out.is_cuda:
        raise TypeError("transpose_to_b16: out must be an int16 CUDA tensor")
    _lib.check(_lib.load().pulse_transpose_to_b16(x.data_ptr() + 4 * x_off, ld_in, rows, cols, out.data_ptr() + 2 * out_off, ld_out, batch, stride_in,
                                                  stride_out, _stream()), "pulse_transpose_to_b16")


def colsum_partial_b16(x, m, n, ld, num_chunks, partial, ld_partial, x_off=0, partial_off=0):
    if x.dtype != torch.int16 or not x.is_cuda:
        raise TypeError("colsum_partial_b16: x must be an int16 (bf16 bit pattern) CUDA tensor")
    _chk(partial, "partial")
    _lib.check(_lib.load().pulse_colsum_partial_b16(x.data_ptr() + 2 * x_off, m, n, ld, num_chunks, partial.data_ptr() + 4 * partial_off, ld_partial,
                                                    _stream()), "pulse_colsum_partial_b16")


def disc_penalty(G, rows, cols, scale, partials, out32=None, out16=None, out32_off=0, out16_off=0, ld32=0, ld16=0):
    """partials[block] = sum G^2; out32 / out16 (+ element offsets) = scale * G as fp32 / bf16."""
    _chk(G, "G"), _chk(partials, "partials"), _chk(out32, "out32")
    if out16 is not None and (out16.dtype != torch.int16 or not out16.is_cuda):
        raise TypeError("disc_penalty: out16 must be an int16 CUDA tensor")
    _lib.check(_lib.load().pulse_disc_penalty(G.data_ptr(), G.stride(0), rows, cols, float(scale),
                                              (out32.data_ptr() + 4 * out32_off) if out32 is not None else None, ld32,
                                              (out16.data_ptr() + 2 * out16_off) if out16 is not None else None, ld16,
                                              partials.data_ptr(), partials.numel(), _stream()), "pulse_disc_penalty")


def disc_reg(flat, grad, ranges, partials):
    """ranges: [(offset, length, alpha)] (<= 4).  grad[off + i] += alpha * flat[off + i]; partials (blocks, 4) = per-block sums of flat[range]^2."""
    _chk(flat, "flat"), _chk(grad, "grad"), _chk(partials, "partials")
    n = len(ranges)
    if not 1 <= n <= 4 or partials.dim() != 2 or partials.shape[1] != 4 or not partials.is_contiguous():
        raise ValueError("disc_reg: 1..4 ranges, partials (blocks, 4) contiguous")
    offs = (ctypes.c_int64 * n)(*[int(r[0]) for r in ranges])
    lens = (ctypes.c_int64 * n)(*[int(r[1]) for r in ranges])
    als = (ctypes.c_float * n)(*[float(r[2]) for r in ranges])
    _lib.check(_lib.load().pulse_disc_reg(flat.data_ptr(), _p(grad), n, offs, lens, als, partials.data_ptr(), partials.shape[0], _stream()), "pulse_disc_reg")


def carve_reduce_regions(base, parts, max_regions=8):
    """Regions of a fused gradient reduce (ReduceGrads).  ``base``: [(offset, count, nslabs)], the slab-summed ranges of the flat gradient in
    order; ``parts``: [(dst_offset, count, rows, src)], ranges whose value is the ordered sum of ``rows`` partial rows of the 2-D tensor ``src``
    (Plan.partial_reduces).  Every part is carved out of the base range that contains it and becomes a region with its own source.
    -> (regions, fused): fused is False -- and the regions are the plain base ranges -- when there is no part, a part straddles base ranges or
    overlaps another, an offset / count / row stride is not a multiple of 4 floats, or more than ``max_regions`` regions would result; the
    caller then keeps the plan's small reduce launches."""
    plain = [(o, c, n, 0.0) for o, c, n in base]
    todo = sorted(parts, key=lambda r: r[0])
    if not todo:
        return plain, False
    regions, used = [], 0
    for off, cnt, ns in base:
        pos, end = off, off + cnt
        for doff, dcnt, rows, src in todo:
            if doff < off or doff >= end:
                continue
            if doff < pos or doff + dcnt > end or doff % 4 or dcnt % 4 or src.stride(0) % 4 or rows < 1:
                return plain, False
            if doff > pos:
                regions.append((pos, doff - pos, ns, 0.0))
            regions.append((doff, dcnt, rows, 0.0, src, src.stride(0)))
            pos = doff + dcnt
            used += 1
        if end > pos:
            regions.append((pos, end - pos, ns, 0.0))
    if used != len(todo) or len(regions) > max_regions:
        return plain, False
    return regions, True


class ReduceGrads:
    """A pre-built pulse_reduce_grads launch: regions = [(offset, count, nslabs, alpha[, src, src_stride])] of a flat gradient buffer; a region
    with ``src`` (a float32 device tensor) sums nslabs rows of that buffer (row stride ``src_stride`` floats) instead of the slabs."""

    def __init__(self, slabs, slab_stride, regions, out, flat=None):
        n = len(regions)
        if not 1 <= n <= 32:
            raise ValueError("ReduceGrads: 1..32 regions")
        _chk(slabs, "slabs"), _chk(out, "out"), _chk(flat, "flat")
        self.n = n
        self.offs = (ctypes.c_int64 * n)(*[int(r[0]) for r in regions])
        self.cnts = (ctypes.c_int64 * n)(*[int(r[1]) for r in regions])
        self.nsl = (ctypes.c_int32 * n)(*[int(r[2]) for r in regions])
        self.als = (ctypes.c_float * n)(*[float(r[3]) for r in regions])
        self._src_keep = [r[4] if len(r) > 4 else None for r in regions]          # (the tensors stay alive with the launch)
        for t in self._src_keep:
            _chk(t, "region src", contiguous=False)
        self.srcs = (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in self._src_keep])
        self.sstr = (ctypes.c_int64 * n)(*[(int(r[5]) if len(r) > 5 and r[4] is not None else 0) for r in regions])
        self.slabs, self.stride, self.out, self.flat = slabs, int(slab_stride), out, flat
        self.has_alpha = any(float(r[3]) != 0.0 for r in regions)

    def run(self, scale=1.0, sq_partials=None, w2_partials=None, num_blocks=1024, alphas=None):
        _chk(sq_partials, "sq_partials"), _chk(w2_partials, "w2_partials")
        if alphas is not None:                       # per-call regulariser coefficients (one per region)
            if len(alphas) != self.n:
                raise ValueError("ReduceGrads: one alpha per region")
            self.als = (ctypes.c_float * self.n)(*[float(a) for a in alphas])
            self.has_alpha = any(float(a) != 0.0 for a in alphas)
        if sq_partials is not None:
            num_blocks = sq_partials.numel()
        if w2_partials is not None and (w2_partials.numel() != 8 * num_blocks or not w2_partials.is_contiguous()):
            raise ValueError("ReduceGrads: w2_partials must be (num_blocks, 8) contiguous, num_blocks = sq_partials.numel() (default 1024)")
        if (w2_partials is not None or self.has_alpha) and self.flat is None:
            raise ValueError("ReduceGrads: regulariser terms need the flat parameter buffer")
        _lib.check(_lib.load().pulse_reduce_grads(self.slabs.data_ptr(), self.stride, self.n, self.offs, self.cnts, self.nsl, self.als, self.srcs, self.sstr,
                                                  self.out.data_ptr(), float(scale), _p(self.flat), _p(sq_partials), _p(w2_partials), num_blocks, _stream()),
                   "pulse_reduce_grads")


def disc_reward(logits, n, scale, out):
    _chk(logits, "logits", contiguous=False), _chk(out, "out", contiguous=False)
    _lib.check(_lib.load().pulse_disc_reward(logits.data_ptr(), logits.stride(0), n, float(scale), out.data_ptr(), out.stride(0), _stream()), "pulse_disc_reward")


def adam_step(params, grads, exp_avg, exp_avg_sq, count, *, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              max_norm=0.0, sqnorm_partials=None, grad_norm_out=None):
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (sqnorm_partials, "sqnorm_partials"),
                  (grad_norm_out, "grad_norm_out")):
        _chk(t, nm)
        if t is not None and nm in ("params", "grads", "exp_avg", "exp_avg_sq") and (t.numel() < count or not t.is_contiguous()):
            raise ValueError(f"adam_step: {nm} must be a contiguous buffer of at least {count} floats")
    _lib.check(_lib.load().pulse_adam_step(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), count, float(lr), beta1, beta2,
                                           eps, weight_decay, int(step), float(max_norm), _p(sqnorm_partials),
                                           sqnorm_partials.numel() if sqnorm_partials is not None else 0, _p(grad_norm_out),
                                           _stream()), "pulse_adam_step")


def adam_step_multi(groups, *, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_norm=0.0, sqnorm_partials=None, grad_norm_out=None):
    """adam_step over several flat buffers -- groups = [(params, grads, exp_avg, exp_avg_sq, count)] -- in ONE launch (pulse_adam_step_multi)."""
    n = len(groups)
    if not 1 <= n <= 4:
        raise ValueError("adam_step_multi: 1..4 groups")
    _chk(sqnorm_partials, "sqnorm_partials"), _chk(grad_norm_out, "grad_norm_out")
    cols = [[], [], [], []]
    for g in groups:
        for k, nm in enumerate(("params", "grads", "exp_avg", "exp_avg_sq")):
            _chk(g[k], nm)
            if g[k].numel() < g[4] or not g[k].is_contiguous():
                raise ValueError(f"adam_step_multi: {nm} must be a contiguous buffer of at least {g[4]} floats")
            cols[k].append(g[k].data_ptr())
    arrs = [(ctypes.c_void_p * n)(*c) for c in cols]
    counts = (ctypes.c_int64 * n)(*[int(g[4]) for g in groups])
    _lib.check(_lib.load().pulse_adam_step_multi(n, arrs[0], arrs[1], arrs[2], arrs[3], counts, float(lr), beta1, beta2, eps, weight_decay, int(step),
                                                 float(max_norm), _p(sqnorm_partials), sqnorm_partials.numel() if sqnorm_partials is not None else 0,
                                                 _p(grad_norm_out), _stream()), "pulse_adam_step_multi")


def rollout_record(*, rewards, dones, terminate, value_raw, value_stride, value_mean, value_var, value_eps, buf_rewards, buf_next_values,
                   buf_dones, env_stride, current_rewards, current_lengths, meter_rewards, meter_lengths, meter_max_size, done_mask,
                   reward_scale=1.0, reward_shift=0.0, buf_terminate=None, meter_partials=None):
    """play_steps bookkeeping of one rollout step in one launch (include/pulse_hip.h section 2c)."""
    a = _lib.RolloutRecordArgs()
    a.num_envs = rewards.numel()
    a.rewards, a.reward_scale, a.reward_shift = _chk(rewards, "rewards"), float(reward_scale), float(reward_shift)
    a.dones, a.terminate = _chk(dones, "dones", torch.int64), _chk(terminate, "terminate", torch.int64)
    _chk(done_mask, "done_mask", torch.uint8), _chk(current_rewards, "current_rewards"), _chk(current_lengths, "current_lengths")
    _chk(meter_rewards, "meter_rewards"), _chk(meter_lengths, "meter_lengths"), _chk(value_mean, "value_mean", torch.float64)
    _chk(value_var, "value_var", torch.float64)
    a.value_raw, a.value_stride = _p(value_raw), int(value_stride)
    a.value_mean, a.value_var, a.value_eps = _p(value_mean), _p(value_var), float(value_eps)
    a.buf_rewards, a.buf_next_values, a.buf_dones, a.env_stride = _p(buf_rewards), _p(buf_next_values), _p(buf_dones), int(env_stride)
    a.current_rewards, a.current_lengths = _p(current_rewards), _p(current_lengths)
    a.meter_rewards, a.meter_lengths, a.meter_max_size = _p(meter_rewards), _p(meter_lengths), float(meter_max_size)
    a.done_mask, a.buf_terminate = _p(done_mask), _p(buf_terminate)
    if meter_partials is not None:          # (blocks, 4) row of this step: the meter updates are applied later by rollout_meters
        _chk(meter_partials, "meter_partials")
        if meter_partials.dim() != 2 or meter_partials.shape[1] != 4 or not meter_partials.is_contiguous():
            raise ValueError("rollout_record: meter_partials must be a contiguous (blocks, 4) tensor")
        a.meter_partials, a.meter_blocks = meter_partials.data_ptr(), meter_partials.shape[0]
    _lib.check(_lib.load().pulse_rollout_record(ctypes.byref(a), _stream()), "pulse_rollout_record")


def rollout_meters(partials, meter_rewards, meter_lengths, meter_max_size):
    """The deferred AverageMeter updates of a rollout: partials (steps, blocks, 4) from rollout_record(meter_partials=partials[step])."""
    _chk(partials, "partials"), _chk(meter_rewards, "meter_rewards"), _chk(meter_lengths, "meter_lengths")
    if partials.dim() != 3 or partials.shape[2] != 4 or not partials.is_contiguous():
        raise ValueError("rollout_meters: partials must be a contiguous (steps, blocks, 4) tensor")
    _lib.check(_lib.load().pulse_rollout_meters(partials.data_ptr(), partials.shape[0], partials.shape[1], meter_rewards.data_ptr(), meter_lengths.data_ptr(),
                                               float(meter_max_size), _stream()), "pulse_rollout_meters")


def kinematic_sim_step(target_rb, noise_rb, rb, target_dof_pos, noise_dof_pos, dof_pos, target_dof_vel, noise_dof_vel, dof_vel, force_src,
                       dof_force):
    n, j = rb.shape[0], rb.shape[1]
    _lib.check(_lib.load().pulse_kinematic_sim_step(_p(target_rb), _p(noise_rb), _p(rb), n, j, _p(target_dof_pos), _p(noise_dof_pos),
                                                    _p(dof_pos), _p(target_dof_vel), _p(noise_dof_vel), _p(dof_vel), _p(force_src),
                                                    _p(dof_force), dof_pos.shape[1], _stream()), "pulse_kinematic_sim_step")

