"""Training path: torch.autograd.Functions over the HIP forward / backward kernels.

The reference trains through torch autograd (opv2v/opencood/tools/train_camera.py:143-179: model.train(), loss.backward(),
optimizer.step(); nuscenes/cross_view_transformer/model/model_module.py:35-60).  Here the attention core (gathered window / dilated-grid
partition, relative-position bias, key mask, softmax, PV, partition reverse; optionally with its log-sum-exp as a differentiable output),
LayerNorm, GELU, swish, every convolution and dense projection (implicit-GEMM forward / input gradient, cobevt_conv_wgrad[_blocked] weight
gradient, operands prepared by csrc/train_prep.hip), the depthwise convolution, BatchNorm (+ residual + ReLU), max-pool, PixelUnshuffle,
nearest / bilinear resizes, the STTF and pairwise warps and the losses run as HIP kernels in both directions - no vendor GEMM, no MIOpen.
The attention / LayerNorm / GELU kernels compute in fp32 (their inputs are cast at the Function boundary under autocast, see _amp_fwd);
the convolution family follows a bf16 autocast region (bf16 storage, fp32 master weights and gradients).

No CPU path: every Function raises on CPU tensors (lib.CobevtHipError) like the inference ops do.

Library variant: every launch below goes to `_L.load("")`, the native library, whatever host.set_compute_dtype selected for INFERENCE
(under "fp32_split" / "fp32_fast" the process-global variant would otherwise route train_rows.hip and attention_bwd.hip
through split products while wgrad3 and the rest stayed exact - a mixed, untested training arithmetic).  The split matrix paths are
inference modes; training is exact fp32 or bf16 autocast.
"""
import ctypes

import torch

from . import lib as _L
from . import ops
from .lib import CobevtHipError

_p, _stream, _ints, _need_cuda = ops._p, ops._stream, ops._ints, ops._need_cuda

# Mixed precision (train_camera.py:123-124,157-160,174-177: `--half` = torch.cuda.amp.autocast + GradScaler): inside an autocast
# region every Function below receives its floating-point inputs cast to fp32 and runs with autocast off - the HIP kernels of
# the training slice compute in fp32, i.e. at or above the precision autocast would have picked - while the torch ops between
# them (the projections, batch norm, the warps) follow the autocast dtype; GradScaler sees ordinary fp32 parameter gradients.
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def _f32c(t, what):
    if t.dtype != torch.float32:
        raise CobevtHipError("%s: the training slice is fp32 (got %s)" % (what, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _attn_dims(batch, heads, ldq, ldk, ldv, ldo, bias_table, bias_L, qmap, kmap, omap):
    L = qmap[6] * qmap[7]
    return _ints([ops.FP32, batch, L, heads, ldq, ldk, ldv, ldo, 0, 0, 0, 0,
                  0 if bias_table is None else 1, 0 if bias_table is None else bias_table.shape[0], bias_L, 0]
                 + list(qmap) + list(kmap) + list(omap))


def _rows_view(t, what):
    """A (rows, d) fp32 matrix whose rows are `ld` floats apart (a column slice of a fused projection is fine)."""
    if t.dim() != 2 or t.dtype != torch.float32 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        raise CobevtHipError("%s must be a 2-D fp32 matrix with unit column stride and 16-byte aligned rows" % what)
    return t.stride(0)


_LN2 = 0.6931471805599453


# ----------------------------------------------------------------------------------------------
# zero-initialised gradient buffers (atomically accumulated weight / bias / bias-table gradients): slices of one zero-filled chunk
# instead of a fill launch each - a 5-agent step made ~490 such tensors, most of them a few KB (profiles/r03_train_amp_kernel_trace.txt).
# A slice is handed out once; a chunk is freed when the tensors cut from it are.  Inside a HIP-graph capture only between
# begin_capture_zero_pool() / end_capture_zero_pool() (host/train_graph.py), so that the chunk's fill is a node of THAT graph.
# ----------------------------------------------------------------------------------------------
USE_ZERO_POOL = True
_ZERO_CHUNK_BYTES = 16 << 20
_ZERO_POOL_MAX_BYTES = 1 << 20        # larger buffers keep their own fill
_ZERO_POOL = {"eager": {}, "capture": None}


def begin_capture_zero_pool():
    _ZERO_POOL["capture"] = {}


def end_capture_zero_pool():
    _ZERO_POOL["capture"] = None


def _zeros(shape, device, dtype):
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    n = 1
    for v in shape:
        n *= v
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    device = torch.device(device)
    if not USE_ZERO_POOL or device.type != "cuda" or nbytes == 0 or nbytes > _ZERO_POOL_MAX_BYTES:
        return torch.zeros(shape, device=device, dtype=dtype)
    if torch.cuda.is_current_stream_capturing():
        pool = _ZERO_POOL["capture"]
        if pool is None:
            return torch.zeros(shape, device=device, dtype=dtype)
    else:
        pool = _ZERO_POOL["eager"]
    # a chunk belongs to the stream it was filled on: a slice handed to work on another stream could be read before the fill ran
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    st = pool.get(key)
    need = (nbytes + 255) // 256 * 256
    if st is None or st[1] + need > st[0].numel():
        st = [torch.zeros(_ZERO_CHUNK_BYTES, device=device, dtype=torch.uint8), 0]
        pool[key] = st
    out = st[0][st[1]:st[1] + nbytes].view(dtype).view(shape)
    st[1] += need
    return out


USE_ATTN_BWD_BF16 = True      # bf16 autocast regions: the attention backward's products on the bf16 matrix path
USE_ATTN_KV2 = True           # ... with two key tiles per workgroup in the dK / dV kernel (False: one, A/B runs)


class WindowAttentionFn(torch.autograd.Function):
    """out = softmax(scale q k^T + bias[rel(q, k)] + mask) v over the gathered windows (cobevt_window_attention_lse /
    cobevt_window_attention_bwd).  q (Rq, d), k, v (Rk, d) token matrices (views with a row stride are accepted), bias_table
    (rows, heads) | None, mask fp32 | None (no gradient).  cfg = (qmap, kmap, omap, batch, heads, scale, bias_L, out_rows,
    drop_p, drop_seed, seed_dev, want_lse, bf16_mm): drop_p > 0 = dropout on the probabilities with the keep mask hashed from drop_seed (+ the
    device word seed_dev, see dropout_step); want_lse: also return the natural log-sum-exp of every query's logits, (batch, windows,
    heads, Nq), as a DIFFERENTIABLE output (its gradient enters the backward kernels as D - dlse)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, bias_table, mask, cfg):
        qmap, kmap, omap, batch, heads, scale, bias_L, out_rows, drop_p, drop_seed, seed_dev, want_lse, _ = cfg
        _need_cuda(q, k, v, bias_table, mask)
        ldq, ldk, ldv = _rows_view(q, "q"), _rows_view(k, "k"), _rows_view(v, "v")
        d = heads * 32
        if q.shape[1] != d or k.shape[1] != d or v.shape[1] != d:
            raise CobevtHipError("window attention: token width must be heads * 32")
        if want_lse and drop_p > 0:
            raise CobevtHipError("window attention: the log-sum-exp output is not available with probability dropout")
        table = None if bias_table is None else _f32c(bias_table, "bias_table")
        mk = None if mask is None else _f32c(mask, "mask")
        L = qmap[6] * qmap[7]
        nq = qmap[1] * qmap[4] * qmap[5]
        out = torch.empty((out_rows, d), device=q.device, dtype=torch.float32)
        lse = torch.empty((batch, L, heads, nq), device=q.device, dtype=torch.float32)
        dims = _attn_dims(batch, heads, ldq, ldk, ldv, d, table, bias_L, qmap, kmap, omap)
        rc = _L.load("").cobevt_window_attention_lse(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(table), _p(mk), dims,
                                                   ctypes.c_float(scale), ctypes.c_float(drop_p), ctypes.c_uint(drop_seed), _p(seed_dev),
                                                   _stream())
        _L.check(rc, "cobevt_window_attention_lse")
        ctx.save_for_backward(q, k, v, out, lse, table, mk)
        ctx.cfg = cfg
        return (out, lse * _LN2) if want_lse else out

    @staticmethod
    @_amp_bwd
    def backward(ctx, dout, dlse=None):
        q, k, v, out, lse, table, mk = ctx.saved_tensors
        qmap, kmap, omap, batch, heads, scale, bias_L, _, drop_p, drop_seed, seed_dev, want_lse, bf16_mm = ctx.cfg
        d = heads * 32
        dout = _f32c(dout, "dout")
        dl = _f32c(dlse, "dlse") if (want_lse and dlse is not None) else None
        # dq is accumulated with atomics by the key tiles of a window; dk / dv rows are written once each
        dq = _zeros((q.shape[0], d), q.device, torch.float32)
        dk = _zeros((k.shape[0], d), q.device, torch.float32)
        dv = _zeros((v.shape[0], d), q.device, torch.float32)
        dbias = None if table is None else _zeros(table.shape, table.device, table.dtype)
        dims = _attn_dims(batch, heads, q.stride(0), k.stride(0), v.stride(0), d, table, bias_L, qmap, kmap, omap)
        # the gradient buffers are dense (ld = d) while q / k / v may be strided views: the kernel shares one ld per
        # operand between the tensor and its gradient, so strided operands are compacted first
        if q.stride(0) != d or k.stride(0) != d or v.stride(0) != d:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            dims = _attn_dims(batch, heads, d, d, d, d, table, bias_L, qmap, kmap, omap)
        if bf16_mm:
            dims[0] |= 0x100 if USE_ATTN_KV2 else 0x300   # the five products on the bf16 matrix path (operands rounded as they are staged; softmax, sums fp32)
        rc = _L.load("").cobevt_window_attention_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), _p(dl), _p(dq), _p(dk), _p(dv),
                                                   _p(dbias), _p(table), _p(mk), dims, ctypes.c_float(scale), ctypes.c_float(drop_p),
                                                   ctypes.c_uint(drop_seed), _p(seed_dev), _stream())
        _L.check(rc, "cobevt_window_attention_bwd")
        return dq, dk, dv, dbias, None, None


class WindowSelfAttentionFn(torch.autograd.Function):
    """WindowAttentionFn on a fused projection: qkv (rows, 3 d) fp32 holds q | k | v side by side (the to_qkv projection of the swap / global
    self-attentions, base_transformer.py:200-237, swap_fusion_modules.py:87-128).  The kernels read the three column blocks in place (row
    stride 3 d) and the backward writes dq | dk | dv into ONE (rows, 3 d) gradient - torch's autograd of three slices is three zero fills,
    three copies and two adds of the full tensor per attention, plus a cast per slice inside autocast regions."""

    @staticmethod
    def forward(ctx, qkv, bias_table, mask, cfg):
        qmap, kmap, omap, batch, heads, scale, bias_L, out_rows, drop_p, drop_seed, seed_dev, want_lse, _ = cfg
        qkv = _f32c(qkv, "qkv")
        _need_cuda(qkv, bias_table, mask)
        d = heads * 32
        if qkv.dim() != 2 or qkv.shape[1] != 3 * d or want_lse:
            raise CobevtHipError("window self-attention: qkv must be (rows, 3 * heads * 32)")
        table = None if bias_table is None else _f32c(bias_table.float(), "bias_table")
        mk = None if mask is None else _f32c(mask, "mask")
        L = qmap[6] * qmap[7]
        nq = qmap[1] * qmap[4] * qmap[5]
        out = torch.empty((out_rows, d), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((batch, L, heads, nq), device=qkv.device, dtype=torch.float32)
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        dims = _attn_dims(batch, heads, 3 * d, 3 * d, 3 * d, d, table, bias_L, qmap, kmap, omap)
        rc = _L.load("").cobevt_window_attention_lse(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(table), _p(mk), dims,
                                                   ctypes.c_float(scale), ctypes.c_float(drop_p), ctypes.c_uint(drop_seed), _p(seed_dev),
                                                   _stream())
        _L.check(rc, "cobevt_window_attention_lse")
        ctx.save_for_backward(qkv, out, lse, table, mk)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, table, mk = ctx.saved_tensors
        qmap, kmap, omap, batch, heads, scale, bias_L, _, drop_p, drop_seed, seed_dev, _, bf16_mm = ctx.cfg
        d = heads * 32
        dout = _f32c(dout.float(), "dout")
        dqkv = _zeros(qkv.shape, qkv.device, torch.float32)
        dbias = None if table is None else _zeros(table.shape, table.device, table.dtype)
        dims = _attn_dims(batch, heads, 3 * d, 3 * d, 3 * d, d, table, bias_L, qmap, kmap, omap)
        if bf16_mm:
            dims[0] |= 0x100 if USE_ATTN_KV2 else 0x300
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        rc = _L.load("").cobevt_window_attention_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), None, _p(dq), _p(dk), _p(dv),
                                                   _p(dbias), _p(table), _p(mk), dims, ctypes.c_float(scale), ctypes.c_float(drop_p),
                                                   ctypes.c_uint(drop_seed), _p(seed_dev), _stream())
        _L.check(rc, "cobevt_window_attention_bwd")
        return dqkv, dbias, None, None


def window_self_attention(qkv, tokmap, batch, heads, scale, out_rows, bias_table=None, bias_L=1, mask=None, drop_p=0.0, drop_seed=None):
    """window_attention(qkv[:, :d], qkv[:, d:2d], qkv[:, 2d:], tokmap, tokmap, tokmap, ...) on the fused (rows, 3 d) projection, one
    gradient tensor back (WindowSelfAttentionFn)"""
    seed_dev = None
    if drop_p > 0 and drop_seed is None:
        drop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        seed_dev = dropout_step(qkv.device)
    mode = _autocast_mode()
    cfg = (tuple(tokmap), tuple(tokmap), tuple(tokmap), int(batch), int(heads), float(scale), int(bias_L), int(out_rows),
           float(drop_p), int(drop_seed or 0), seed_dev, False, bool(USE_ATTN_BWD_BF16 and mode == "bf16"))
    if mode is None:
        return WindowSelfAttentionFn.apply(qkv, bias_table, mask, cfg)
    half = _half_attention_ok((qkv,), drop_p, False, heads) and qkv.shape[1] == 3 * heads * 32
    with torch.autocast("cuda", enabled=False):
        if half:
            return WindowAttentionHalfFn.apply(qkv, None, None, bias_table, mask, cfg, True)
        return WindowSelfAttentionFn.apply(qkv.float(), bias_table, None if mask is None else mask.float(), cfg)


class WindowAttentionHalfFn(torch.autograd.Function):
    """WindowAttentionFn inside a bf16 autocast region, on the projections' own bf16 tensors: no casts at the boundary.  Forward: the bf16
    gather kernel with its log-sum-exp output (bf16 MFMAs, fp32 softmax - what torch's autocast does with the two einsums and the softmax,
    train_camera.py:157-160), out in bf16; backward: the bf16-matrix-path kernels reading q / k / v / out / dout and writing dq / dk / dv
    in bf16 (fp32 accumulation; the bias-table gradient stays fp32).  q, k, v: (rows, d) bf16 with unit column stride - column blocks of
    one fused (rows, 3 d) projection when `fused` (then ONE (rows, 3 d) gradient is returned for it), else three tensors."""

    @staticmethod
    def forward(ctx, q, k, v, bias_table, mask, cfg, fused):
        qmap, kmap, omap, batch, heads, scale, bias_L, out_rows, drop_p, drop_seed, seed_dev, want_lse, _ = cfg
        d = heads * 32
        qkv = None
        if fused:
            qkv = q if q.is_contiguous() else q.contiguous()
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        _need_cuda(q, k, v, bias_table, mask)
        for t in (q, k, v):
            if t.dim() != 2 or t.dtype != torch.bfloat16 or t.shape[1] != d or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
                raise CobevtHipError("bf16 window attention: (rows, heads * 32) bf16 matrices with unit column stride, 16-byte aligned rows")
        if want_lse or drop_p > 0:
            raise CobevtHipError("bf16 window attention: no log-sum-exp output, no probability dropout")
        table = None if bias_table is None else _f32c(bias_table.float(), "bias_table")
        mk = None if mask is None else _f32c(mask.float(), "mask")
        L = qmap[6] * qmap[7]
        nq = qmap[1] * qmap[4] * qmap[5]
        out = torch.empty((out_rows, d), device=q.device, dtype=torch.bfloat16)
        # [0]: the log-sum-exp; [1]: scratch of the backward (D = rowsum(dO o O): the dQ kernel writes it, the dK / dV kernel reads it)
        lse = torch.empty((2, batch, L, heads, nq), device=q.device, dtype=torch.float32)
        dims = _attn_dims(batch, heads, q.stride(0), k.stride(0), v.stride(0), d, table, bias_L, qmap, kmap, omap)
        dims[0] = ops.BF16
        rc = _L.load("").cobevt_window_attention_lse(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(table), _p(mk), dims,
                                                   ctypes.c_float(scale), ctypes.c_float(0.0), ctypes.c_uint(0), None, _stream())
        _L.check(rc, "cobevt_window_attention_lse")
        if fused:
            ctx.save_for_backward(qkv, out, lse, table, mk)
        else:
            ctx.save_for_backward(q, k, v, out, lse, table, mk)
        ctx.cfg = cfg
        ctx.fused = bool(fused)
        return out

    @staticmethod
    def backward(ctx, dout):
        qmap, kmap, omap, batch, heads, scale, bias_L, _, _, _, _, _, _ = ctx.cfg
        d = heads * 32
        if ctx.fused:
            qkv, out, lse, table, mk = ctx.saved_tensors
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            dqkv = torch.zeros_like(qkv)
            dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        else:
            q, k, v, out, lse, table, mk = ctx.saved_tensors
            # the kernels share one row stride per operand between a tensor and its gradient
            dq, dk, dv = (torch.zeros((t.shape[0], t.stride(0)), device=t.device, dtype=torch.bfloat16)[:, :d] if t.stride(0) != d
                          else torch.zeros_like(t) for t in (q, k, v))
        dout = dout.to(torch.bfloat16)
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dbias = None if table is None else _zeros(table.shape, table.device, table.dtype)
        dims = _attn_dims(batch, heads, q.stride(0), k.stride(0), v.stride(0), d, table, bias_L, qmap, kmap, omap)
        dims[0] = ops.BF16 | 0x100 | (0x400 if USE_ATTN_D_SCRATCH else 0)
        rc = _L.load("").cobevt_window_attention_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), None, _p(dq), _p(dk), _p(dv),
                                                   _p(dbias), _p(table), _p(mk), dims, ctypes.c_float(scale), ctypes.c_float(0.0),
                                                   ctypes.c_uint(0), None, _stream())
        _L.check(rc, "cobevt_window_attention_bwd")
        if ctx.fused:
            return dqkv, None, None, dbias, None, None, None
        return dq, dk, dv, dbias, None, None, None


USE_ATTN_D_SCRATCH = True     # ... and D = rowsum(dO o O) handed from the dQ kernel to the dK / dV kernel through the lse tensor's second half
USE_ATTN_HALF_IO = True       # bf16 autocast regions: bf16 q / k / v / out / gradients through the attention Functions (no boundary casts)


def _half_attention_ok(tensors, drop_p, return_lse, heads):
    d = heads * 32
    return (USE_ATTN_HALF_IO and USE_ATTN_BWD_BF16 and USE_ATTN_KV2 and _autocast_mode() == "bf16" and drop_p == 0 and not return_lse
            and all(t.dtype == torch.bfloat16 and t.dim() == 2 and t.shape[1] in (d, 3 * d) and t.stride(1) == 1 and t.stride(0) % 8 == 0
                    and t.data_ptr() % 16 == 0 for t in tensors))


_DROPOUT_STEP = {}


def dropout_step(device):
    """The per-device int32 word the attention-probability dropout adds to its seed.  Eager training does not need it (a fresh host
    seed is drawn per call); a training step replayed from a captured HIP graph does `dropout_step(dev).add_(1)` INSIDE the graph,
    because the host seeds are frozen into the graph's kernel nodes at capture (tools/train_graph_probe.py)."""
    key = str(torch.device(device))
    if key not in _DROPOUT_STEP:
        _DROPOUT_STEP[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _DROPOUT_STEP[key]


def window_attention(q, k, v, qmap, kmap, omap, batch, heads, scale, out_rows, bias_table=None, bias_L=1, mask=None, drop_p=0.0,
                     drop_seed=None, return_lse=False):
    """drop_p > 0: dropout on the attention probabilities.  drop_seed None draws a host seed from torch's CPU generator (so
    torch.manual_seed makes a run repeatable, and no device round trip is needed) and adds the device word dropout_step();
    an explicit drop_seed is used as it is (tests: attention_dropout_mask reproduces the mask)"""
    seed_dev = None
    if drop_p > 0 and drop_seed is None:
        drop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        seed_dev = dropout_step(q.device)
    # inside a bf16 autocast region torch runs the two einsums of the attention (and their gradients) in bf16 and the softmax in fp32
    # (train_camera.py:157-160).  q / k / v arrive as bf16 projections, so the fp32 forward kernel already sees bf16-representable operands;
    # the backward kernels take the bf16 matrix path (16 x fewer matrix cycles) with dO, P and dZ rounded like autocast's backward does
    bf16_mm = USE_ATTN_BWD_BF16 and _autocast_mode() == "bf16"
    cfg = (tuple(qmap), tuple(kmap), tuple(omap), int(batch), int(heads), float(scale), int(bias_L), int(out_rows),
           float(drop_p), int(drop_seed or 0), seed_dev, bool(return_lse), bool(bf16_mm))
    if _half_attention_ok((q, k, v), drop_p, return_lse, heads) and q.shape[1] == heads * 32:
        with torch.autocast("cuda", enabled=False):
            return WindowAttentionHalfFn.apply(q, k, v, bias_table, mask, cfg, False)
    return WindowAttentionFn.apply(q, k, v, bias_table, mask, cfg)


def attention_dropout_mask(batch, windows, heads, nq, nk, drop_p, drop_seed, device):
    """bool (batch, windows, heads, nq, nk): the keep mask the training kernels use for (drop_p, drop_seed) (test hook)"""
    keep = torch.empty((batch, windows, heads, nq, nk), device=device, dtype=torch.uint8)
    _need_cuda(keep)
    rc = _L.load("").cobevt_attention_dropout_mask(batch, windows, heads, nq, nk, ctypes.c_float(drop_p), ctypes.c_uint(drop_seed),
                                                 _p(keep), _stream())
    _L.check(rc, "cobevt_attention_dropout_mask")
    return keep.bool()


def _autocast_mode():
    """None outside autocast regions, "bf16" inside a bfloat16 one, "other" (fp16: the kernels' fp32 forms) otherwise"""
    if not torch.is_autocast_enabled():
        return None
    return "bf16" if torch.get_autocast_dtype("cuda") == torch.bfloat16 else "other"


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension.  fp32 in / out: cobevt_layernorm / cobevt_layernorm_bwd.  Inside a bf16 autocast region
    (layernorm() below) x may be bf16 (a projection's output) and the result is written as bf16 when its consumer is a projection
    (`out_bf16`): torch's autocast computes layer_norm in fp32 and casts its result for the linear that follows (train_camera.py:157-160) -
    cobevt_layernorm_fwd_t does that rounding in the same pass, and cobevt_layernorm_bwd_t reads the bf16 gradient the projection's input
    gradient produced; statistics, arithmetic and the parameter gradients are fp32 in every case."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_bf16):
        _need_cuda(x, gamma, beta)
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise CobevtHipError("layernorm: fp32 or bf16 input (got %s)" % x.dtype)
        x = x if x.is_contiguous() else x.contiguous()
        g, b = _f32c(gamma, "gamma"), _f32c(beta, "beta")
        if x.dtype == torch.float32 and not out_bf16:
            y = ops.layernorm(x, g, b, eps)
        else:
            C = x.shape[-1]
            y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
            rc = _L.load("").cobevt_layernorm_fwd_t(_p(x), _p(g), _p(b), _p(y), x.numel() // C, C, ctypes.c_float(eps),
                                                  _ints([ops.dcode(x.dtype), ops.dcode(y.dtype)]), _stream())
            _L.check(rc, "cobevt_layernorm_fwd_t")
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dy = dy if dy.is_contiguous() else dy.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dg = _zeros(C, x.device, torch.float32)
        db = _zeros(C, x.device, torch.float32)
        if x.dtype == torch.float32 and dy.dtype == torch.float32:
            rc = _L.load("").cobevt_layernorm_bwd(_p(x), _p(dy), _p(g), _p(dx), _p(dg), _p(db), rows, C, ctypes.c_float(ctx.eps), _stream())
            _L.check(rc, "cobevt_layernorm_bwd")
        else:
            rc = _L.load("").cobevt_layernorm_bwd_t(_p(x), _p(dy), _p(g), _p(dx), _p(dg), _p(db), rows, C, ctypes.c_float(ctx.eps),
                                                  _ints([ops.dcode(x.dtype), ops.dcode(dy.dtype), ops.dcode(dx.dtype)]), _stream())
            _L.check(rc, "cobevt_layernorm_bwd_t")
        return dx, dg, db, None, None


def layernorm(x, ln, for_projection=False):
    """x through the nn.LayerNorm container `ln`.  for_projection: the only consumer is linear() / linear_weight() - inside a bf16 autocast
    region the result is then written in bf16, the operand type of that projection (what torch's autocast cast produces, minus the cast)."""
    mode = _autocast_mode()
    if mode is None:
        return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, False)
    with torch.autocast("cuda", enabled=False):
        if mode != "bf16" or x.dtype not in (torch.float32, torch.bfloat16):
            return LayerNormFn.apply(x.float(), ln.weight.float(), ln.bias.float(), ln.eps, False)
        return LayerNormFn.apply(x, ln.weight.float(), ln.bias.float(), ln.eps, bool(for_projection))


class GeluFn(torch.autograd.Function):
    """nn.GELU() (exact erf form): cobevt_gelu in both directions; bf16 tensors (a projection's output inside a bf16 autocast region - torch
    runs gelu in its input's type) on cobevt_gelu_bf16: bf16 storage, fp32 arithmetic."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x if x.is_contiguous() else x.contiguous()
        y = torch.empty_like(x)
        if x.dtype == torch.bfloat16:
            rc = _L.load("").cobevt_gelu_bf16(_p(x), None, _p(y), x.numel(), _stream())
            _L.check(rc, "cobevt_gelu_bf16")
        else:
            x = _f32c(x, "gelu input")
            rc = _L.load("").cobevt_gelu(_p(x), None, _p(y), x.numel(), _stream())
            _L.check(rc, "cobevt_gelu")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.to(x.dtype)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(x)
        if x.dtype == torch.bfloat16:
            rc = _L.load("").cobevt_gelu_bf16(_p(x), _p(dy), _p(dx), x.numel(), _stream())
            _L.check(rc, "cobevt_gelu_bf16")
        else:
            rc = _L.load("").cobevt_gelu(_p(x), _p(dy), _p(dx), x.numel(), _stream())
            _L.check(rc, "cobevt_gelu")
        return dx


def gelu(x):
    mode = _autocast_mode()
    if mode is None:
        return GeluFn.apply(x)
    with torch.autocast("cuda", enabled=False):
        if mode == "bf16" and x.dtype == torch.bfloat16 and x.numel() % 8 == 0:
            return GeluFn.apply(x)
        return GeluFn.apply(x.float())


_SCRATCH_BLOCKS = 512         # per-workgroup partial sums of the channel reductions (fp64 [blocks][2][C]); see csrc/train_glue.hip


def _scratch(c, device):
    return torch.empty(_SCRATCH_BLOCKS * 2 * c, device=device, dtype=torch.float64)


def column_sum(rows2d):
    """fp32 (C,) = sum over the rows of a contiguous (rows, C) fp32 / bf16 matrix (bias gradients): cobevt_channel_sums, fp64
    partial sums per workgroup"""
    _need_cuda(rows2d)
    m, c = rows2d.shape
    acc = torch.empty(c, device=rows2d.device, dtype=torch.float64)
    lib = _L.load("")
    if c % 8:
        out = torch.empty(c, device=rows2d.device, dtype=torch.float32)
        _L.check(lib.cobevt_channel_sums(_p(rows2d), None, _p(acc), None, None, None, 0, ops.dcode(rows2d.dtype), m, c, _stream()),
                 "cobevt_channel_sums")
        _L.check(lib.cobevt_f64_to_f32(_p(acc), _p(out), c, _stream()), "cobevt_f64_to_f32")
        return out
    out = torch.empty(c, device=rows2d.device, dtype=torch.float32)
    _L.check(lib.cobevt_channel_sums(_p(rows2d), None, _p(acc), None, _p(out), _p(_scratch(c, rows2d.device)), _SCRATCH_BLOCKS,
                                     ops.dcode(rows2d.dtype), m, c, _stream()), "cobevt_channel_sums")
    return out


USE_LIBRARY_GEMM = False     # True: nn.Linear through torch / rocBLAS (A/B runs); default: the package's own kernels


def linear(x, lin):
    """nn.Linear container on (..., K) rows.  A dense projection is a 1 x 1 convolution over a (1, rows, 1, K) channels-last map:
    forward and input gradient on the implicit-GEMM kernel (csrc/igemm.hip), weight gradient on cobevt_conv_wgrad (a GEMM over the
    rows), bias gradient a column sum - no vendor GEMM (train_camera.py:143-179 runs these through cuBLAS)."""
    return linear_weight(x, lin.weight, lin.bias)


def linear_weight(x, weight, bias=None):
    """linear() on a (out, K) weight tensor (a parameter, or a differentiable function of one - a reshaped / zero-padded 1 x 1 conv weight)"""
    _need_cuda(x)
    if USE_LIBRARY_GEMM:                           # explicit A/B switch only (tools/train_grad_diag.py): never taken by shape
        return torch.nn.functional.linear(x, weight, bias)
    if x.shape[-1] % 4 or weight.shape[0] % 4:
        # widths off the kernels' 16-byte pieces: zero-pad K and / or the output width to the next multiple of 4 (memory ops with
        # exact adjoints - zero columns add nothing to a product) and run the SAME kernels; no library GEMM behind any shape
        k, n = x.shape[-1], weight.shape[0]
        kp, npad = -(-k // 4) * 4, -(-n // 4) * 4
        F = torch.nn.functional
        y = linear_weight(F.pad(x, (0, kp - k)), F.pad(weight, (0, kp - k, 0, npad - n)), None if bias is None else F.pad(bias, (0, npad - n)))
        return y[..., :n]
    k = x.shape[-1]
    rows = x.numel() // k
    # any (H, W) factorisation of the rows is the same 1 x 1 convolution; the blocked weight-gradient kernel walks 16-pixel blocks of
    # map rows and shares rows out over workgroups, so the rows are presented as an H x W map with a wide W (as (rows, 1) every
    # "map row" was one pixel padded to 16: 20 ms of padding copies per 5-agent step)
    mh, mw = rows, 1
    for cand in (256, 128, 64, 32, 16):
        if rows % cand == 0 and rows // cand >= 4:
            mh, mw = rows // cand, cand
            break
    x4 = x.reshape(1, mh, mw, k).permute(0, 3, 1, 2)                   # (1, K, H, W)-shaped view of channels-last memory
    w4 = _master(weight)[:, :, None, None]
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        with torch.autocast("cuda", enabled=False):
            y = Conv2dFn.apply(x4.to(torch.bfloat16), w4, bias, 1, 0)
    else:
        y = Conv2dFn.apply(x4, w4, bias, 1, 0)
    return y.permute(0, 2, 3, 1).reshape(x.shape[:-1] + (weight.shape[0],))


def _master(weight):
    """The fp32 master copy the training convolutions take (a module kept in fp32 - the reference's --half loop keeps fp32 parameters
    under autocast - hands its parameter over as it is; anything else goes through a differentiable cast)"""
    return weight if weight.dtype == torch.float32 else weight.float()


def dropout(x, p):
    """nn.Dropout in train mode (elementwise mask; torch's generator so seeds behave like the reference's)."""
    return torch.nn.functional.dropout(x, p, True) if p > 0 else x


# ----------------------------------------------------------------------------------------------
# convolutions (the convolutional half of the training path)
# ----------------------------------------------------------------------------------------------
_KLUT = {}


def _weight_rows(weight):
    """(Cout, Cin, kh, kw) -> ([Cout][Kpad] rows with k = (kh * Kw + kw) * Cin + c, K, Kpad) in the weight's dtype: the implicit-GEMM
    kernel's weight layout, produced ON THE DEVICE (a permute + pad) so that it follows the parameter through optimizer steps."""
    cout, cin, kh, kw = weight.shape
    K = kh * kw * cin
    tile = 32 if weight.dtype == torch.bfloat16 else 16
    kpad = (K + tile - 1) // tile * tile
    w2 = weight.permute(0, 2, 3, 1).reshape(cout, K)
    if kpad != K:
        w2 = torch.nn.functional.pad(w2, (0, kpad - K))
    return w2.contiguous(), K, kpad


def _klut(kh, kw, cin, kpad, device):
    key = (kh, kw, cin, kpad, device)
    if key not in _KLUT:
        k = torch.arange(kpad)
        tap, c = k // cin, k % cin
        code = ((tap // kw) << 20) | ((tap % kw) << 10) | c
        code[k >= kh * kw * cin] = -1
        _KLUT[key] = code.to(torch.int32).to(device).contiguous()
    return _KLUT[key]


def _kpad(K, dtype):
    tile = 32 if dtype == torch.bfloat16 else 16
    return (K + tile - 1) // tile * tile


USE_TORCH_OPERAND_PREP = False      # True: weight rows / blocked operands through the torch-op specifications (A/B and diagnostics only)


def conv_weight_rows(weight, dtype, fwd=True, dgrad=False):
    """fp32 master weight (Cout, Cin, kh, kw) -> (rows_fwd [Cout][Kpad] | None, rows_dgrad [Cin][Kpad'] | None) in `dtype`: ONE launch of
    cobevt_conv_weight_rows (csrc/train_prep.hip) instead of cast + permute + pad (+ flip + transpose + permute + pad) per convolution and
    step.  Specification: _weight_rows(weight.to(dtype)) and _weight_rows(weight.to(dtype).flip(2, 3).transpose(0, 1))."""
    cout, cin, kh, kw = weight.shape
    if USE_TORCH_OPERAND_PREP:
        wd = weight.detach().to(dtype)
        return (_weight_rows(wd)[0] if fwd else None), (_weight_rows(wd.flip(2, 3).transpose(0, 1))[0] if dgrad else None)
    w = _f32c(weight.detach(), "weight")
    kpf, kpd = _kpad(kh * kw * cin, dtype), _kpad(kh * kw * cout, dtype)
    rf = torch.empty((cout, kpf), device=w.device, dtype=dtype) if fwd else None
    rd = torch.empty((cin, kpd), device=w.device, dtype=dtype) if dgrad else None
    dims = _ints([ops.dcode(dtype), cout, cin, kh, kw, kpf, kpd])
    _L.check(_L.load("").cobevt_conv_weight_rows(_p(w), _p(rf), _p(rd), dims, _stream()), "cobevt_conv_weight_rows")
    return rf, rd


def _igemm_rows(x, w2, cout, cin, kh, kw, bias, stride, pad, ho, wo):
    """x (N, H, W, Cin) channels-last, w2 = [Cout][Kpad] weight rows (k = (r * kw + s) * Cin + c) of the same dtype (fp32 / bf16) ->
    (N, ho, wo, Cout) in that dtype: cobevt_conv2d_nhwc (csrc/igemm.hip), top / left padding `pad`; taps past the bottom / right edge
    read zeros (the kernel bounds-checks every tap).  Channel counts that are not a multiple of a 16-byte chunk take the kernel's
    gather path, which reads its input as fp32.  bias: fp32 (Cout,) | None."""
    n, h, w, _ = x.shape
    bf16 = x.dtype == torch.bfloat16
    K, kpad = kh * kw * cin, w2.shape[1]
    smallc = int(cin % (8 if bf16 else 4) != 0)
    klut = _klut(kh, kw, cin, kpad, x.device) if smallc else None
    if smallc and bf16:
        x = x.float()
    out = torch.empty((n, ho, wo, cout), device=x.device, dtype=w2.dtype)
    dims = _ints([ops.BF16 if bf16 else ops.FP32, n, h, w, cin, ho, wo, cout, kh, kw, stride, pad, K, kpad, 0, 0, 0, 0, ho, wo, smallc])
    b = None if bias is None else _f32c(bias.detach().float(), "bias")
    rc = _L.load("").cobevt_conv2d_nhwc(_p(x), _p(w2), _p(b), None, None, None, _p(klut), _p(out), dims, _stream())
    _L.check(rc, "cobevt_conv2d_nhwc")
    return out


USE_WGRAD3 = True             # bf16 3x3 / stride-1 weight gradients straight from the channels-last maps (csrc/wgrad3.hip)
USE_TRAIN_STRIPS = True       # bf16 3x3 / pad-1 convolutions with 64 | Cin: forward and (stride 1) input gradient on the inference kernels


def conv3_strips_plan(n, ho, wo, cin, cout, stride):
    """tile choice of ops.conv2d for one bf16 3x3 launch: 0 = the LDS-staged kernel (reads `rows3`), else a strip variant (`frag`)"""
    variant = ops.CONV3_VARIANT or ops.conv3_tiling(n, ho, wo, cin, cout, 64, stride=stride, bf16=True)
    if variant == 0 and stride == 2:
        variant = 151 if cout <= 64 else 150
    return variant


def conv3_weight_operand(weight, variant, dgrad):
    """fp32 master weight (Cout, Cin, 3, 3) -> the bf16 operand that `variant` reads (csrc/train_prep.hip), for the forward convolution or
    (dgrad) for its input gradient = the convolution with the channel roles swapped and the taps flipped.  Specification:
    ops.ConvPlan(weight or weight.flip(2, 3).transpose(0, 1), ...).wfrag / .wgt3."""
    cout, cin = weight.shape[:2]
    o, i = (cin, cout) if dgrad else (cout, cin)
    w = _f32c(weight.detach(), "weight")
    if variant:
        op = torch.empty(((o + 127) // 128 * 4, i // 64, 9, 4, 64, 8), device=w.device, dtype=torch.bfloat16)
    else:
        op = torch.empty((o, i // 64, 9, 64), device=w.device, dtype=torch.bfloat16)
    _L.check(_L.load("").cobevt_conv3_weight_operands(_p(w), _p(op) if variant else None, None if variant else _p(op),
                                                    _ints([cout, cin, int(dgrad)]), _stream()), "cobevt_conv3_weight_operands")
    return op


def conv3_weight_operand_pair(weight, var_f, var_d):
    """conv3_weight_operand(weight, var_f, False) and (var_d >= 0) conv3_weight_operand(weight, var_d, True) from ONE launch"""
    cout, cin = weight.shape[:2]
    w = _f32c(weight.detach(), "weight")

    def buf(o, i, variant):
        if variant:
            return torch.empty(((o + 127) // 128 * 4, i // 64, 9, 4, 64, 8), device=w.device, dtype=torch.bfloat16)
        return torch.empty((o, i // 64, 9, 64), device=w.device, dtype=torch.bfloat16)
    f = buf(cout, cin, var_f)
    d = buf(cin, cout, var_d) if var_d >= 0 else None
    outs = (ctypes.c_void_p * 4)(f.data_ptr() if var_f else None, None if var_f else f.data_ptr(),
                                 d.data_ptr() if (d is not None and var_d) else None, d.data_ptr() if (d is not None and not var_d) else None)
    _L.check(_L.load("").cobevt_conv3_weight_operands2(_p(w), outs, _ints([cout, cin]), _stream()), "cobevt_conv3_weight_operands2")
    return f, d


def _conv3_strips(x, operand, variant, cout, bias, stride):
    """x (N, H, W, Cin) bf16 channels-last -> (N, Ho, Wo, Cout) bf16: cobevt_conv3x3_wfrag_nhwc / cobevt_conv3x3_nhwc (csrc/conv3x3.hip) with
    the operand conv3_weight_operand made for `variant`; bias fp32 (Cout,) | None"""
    n, h, w, cin = x.shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.bfloat16)
    b = None if bias is None else _f32c(bias.detach().float(), "bias")
    if variant:
        dims = _ints([ops.BF16, n, h, w, cin, cout, 0, 0, 0, 64, (cout + 127) // 128 * 128, variant, stride])
        rc = _L.load("").cobevt_conv3x3_wfrag_nhwc(_p(x), _p(operand), _p(b), None, _p(out), dims, _stream())
        _L.check(rc, "cobevt_conv3x3_wfrag_nhwc")
    else:
        dims = _ints([ops.BF16, n, h, w, cin, cout, 0, 0, 0, 64])
        rc = _L.load("").cobevt_conv3x3_nhwc(_p(x), _p(operand), _p(b), None, _p(out), dims, _stream())
        _L.check(rc, "cobevt_conv3x3_nhwc")
    return out


USE_TRAIN_ROWS_GEMM = True    # bf16 dense projections / 1x1 stride-1 convolutions with K, N <= 512: forward and input gradient on the inference row GEMM


def linear_weight_frags(weight2d, forward=True, transposed=False):
    """fp32 master weight (N, K) -> (table for y = x W^T | None, table for dx = dy W | None): the bf16 fragment tables
    cobevt_linear_rows_small_k reads, both from one launch (csrc/train_prep.hip).  Specification: ops.ConvPlan(weight or weight.t(), ...).wfrag_rows."""
    n, k = weight2d.shape
    w = _f32c(weight2d.detach(), "weight")
    np_, kp = (n + 127) // 128 * 128, (k + 127) // 128 * 128
    f = torch.empty((np_ // 32, kp // 16, 64, 8), device=w.device, dtype=torch.bfloat16) if forward else None
    t = torch.empty((kp // 32, np_ // 16, 64, 8), device=w.device, dtype=torch.bfloat16) if transposed else None
    _L.check(_L.load("").cobevt_linear_weight_frags(_p(w), _p(f), _p(t), _ints([n, k]), _stream()), "cobevt_linear_weight_frags")
    return f, t


def _rows_gemm(x2d, frag, n_out, k_in, bias):
    """x2d (M, k_in) bf16 rows -> (M, n_out) bf16 = x W^T (+ bias): cobevt_linear_rows_small_k with the table of linear_weight_frags"""
    m = x2d.shape[0]
    out = torch.empty((m, n_out), device=x2d.device, dtype=torch.bfloat16)
    b = None if bias is None else _f32c(bias.detach().float(), "bias")
    d3 = (ctypes.c_long * 14)(ops.BF16, m, n_out, k_in, k_in, 0, 0, 0, 1, m, 1, m, 1, 32)
    rc = _L.load("").cobevt_linear_rows_small_k(_p(x2d), _p(frag), _p(b), None, None, None, _p(out), d3, ctypes.c_float(0.0), _stream())
    _L.check(rc, "cobevt_linear_rows_small_k")
    return out


USE_WGRAD_BLOCKED = True      # bf16: weight gradient on the bf16 matrix path (cobevt_conv_wgrad_blocked) where wgrad_blocked_mode() has a form for it
USE_WGRAD_BLOCKED_STRIDED = True   # ... incl. the stride-2 and stem forms (modes 1 / 2); False: those stay on cobevt_conv_wgrad (A/B runs)


def _blocked_operands(xl, dyl, k, pad):
    """The operands of cobevt_conv_wgrad_blocked from channels-last bf16 x (N, H, W, Cin) and dy (N, Ho, Wo, Cout): 8 pixels of one
    channel per 16-byte piece, [n][row][block][channel][8]; x zero-padded by `pad` on every side, both rows padded with zero blocks
    (dy to an even number of blocks, x to one block more).  One pad + one permuting copy per operand."""
    n, h, w, cin = xl.shape
    _, ho, wo, cout = dyl.shape
    ndb = (wo + 15) // 16 * 2
    nxb = ndb + 1
    hp = max(h + 2 * pad, ho + k - 1)
    xp = torch.nn.functional.pad(xl, (0, 0, pad, nxb * 8 - w - pad, pad, hp - h - pad))
    dp = torch.nn.functional.pad(dyl, (0, 0, 0, ndb * 8 - wo))
    xb = xp.view(n, hp, nxb, 8, cin).permute(0, 1, 2, 4, 3).contiguous()
    db = dp.view(n, ho, ndb, 8, cout).permute(0, 1, 2, 4, 3).contiguous()
    return xb, db, hp, nxb, ndb


def wgrad_blocked_mode(k, stride, pad, cin):
    """Which form of cobevt_conv_wgrad_blocked serves a bf16 convolution: 0 stride 1 (k = 1 / 3), 1 stride 2 with the columns
    de-interleaved into planes (3x3 / pad 1 and the 1x1 / pad 0 shortcut of a down-sampling BasicBlock), 2 the k tap columns as
    pseudo-channels (the 7x7 / stride 2 stem on 3 channels); None = the generic fp32-matrix kernel (cobevt_conv_wgrad)"""
    if stride == 1 and k in (1, 3):
        return 0
    if stride == 2 and ((k == 3 and pad == 1) or (k == 1 and pad == 0)):
        return 1
    if k in (1, 3, 5, 7) and k * cin <= 32:
        return 2
    return None


def _blocked_geometry(h, w, ho, wo, k, pad, stride, mode):
    ndb = (wo + 15) // 16 * 2
    if mode == 0:
        return ndb, ndb + 1, max(h + 2 * pad, ho + k - 1), 1, 1
    if mode == 1:
        return ndb, ndb + (1 if k == 3 else 0), max(h + 2 * pad, (ho - 1) * 2 + k), (2 if k == 3 else 1), 2
    return ndb, ndb, max(h + 2 * pad, (ho - 1) * stride + k), k, stride


def _blocked_x_general(xl, hp, nxb, pad, planes, sx):
    """torch-op specification of cobevt_wgrad_block_operand for any (planes, sx): [n][row][block][plane][c][8], slot j of plane q of
    block b = zero-padded input pixel sx (8 b + j) + q - pad of input row (row - pad)"""
    n, h, w, c = xl.shape
    width = (nxb * 8 - 1) * sx + planes                                  # padded columns the planes read
    xp = torch.nn.functional.pad(xl, (0, 0, pad, max(0, width - w - pad), pad, hp - h - pad))
    cols = (torch.arange(nxb * 8, device=xl.device) * sx)[:, None] + torch.arange(planes, device=xl.device)[None]     # [pixel slot][plane]
    g = xp[:, :, cols.reshape(-1)]                                         # (n, hp, nxb * 8 * planes, c)
    return g.view(n, hp, nxb, 8, planes, c).permute(0, 1, 2, 4, 5, 3).contiguous()


def blocked_operands(xl, dyl, k, pad, stride=1, mode=0):
    """The operands of cobevt_conv_wgrad_blocked (mode: wgrad_blocked_mode) as two launches of cobevt_wgrad_block_operand - instead of
    two pads + two permuting copies (6-8 launches; _blocked_operands, the mode-0 specification).  Returns (xb, db, hp, nxb, ndb)."""
    n, h, w, cin = xl.shape
    _, ho, wo, cout = dyl.shape
    if USE_TORCH_OPERAND_PREP and mode == 0:
        return _blocked_operands(xl, dyl, k, pad)
    ndb, nxb, hp, planes, sx = _blocked_geometry(h, w, ho, wo, k, pad, stride, mode)
    if USE_TORCH_OPERAND_PREP:
        xb = _blocked_x_general(xl, hp, nxb, pad, planes, sx)
        db = _blocked_x_general(dyl, ho, ndb, 0, 1, 1)
        return xb, db, hp, nxb, ndb
    xb = torch.empty((n, hp, nxb, planes, cin, 8), device=xl.device, dtype=xl.dtype)
    db = torch.empty((n, ho, ndb, 1, cout, 8), device=xl.device, dtype=xl.dtype)
    lib = _L.load("")
    _L.check(lib.cobevt_wgrad_block_operand(_p(xl), _p(xb), _ints([n, h, w, cin, hp, nxb, pad, pad, planes, sx]), _stream()),
             "cobevt_wgrad_block_operand")
    _L.check(lib.cobevt_wgrad_block_operand(_p(dyl), _p(db), _ints([n, ho, wo, cout, ho, ndb, 0, 0, 1, 1]), _stream()),
             "cobevt_wgrad_block_operand")
    return xb, db, hp, nxb, ndb


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (square kernel, symmetric padding, groups 1) on (N, C, H, W)-shaped tensors (channels-last memory is used as it
    is).  Forward and the input gradient run on the implicit-GEMM kernel (the input gradient is the same convolution with the
    taps flipped and the channel roles swapped, on the zero-stuffed output gradient when stride > 1); the weight gradient is
    cobevt_conv_wgrad (csrc/train_rows.hip: a GEMM over the pixels on the fp32 matrix path, operands straight from global memory)
    or cobevt_conv_wgrad_blocked (bf16 matrix path).  weight / bias are the fp32 MASTER parameters; x is fp32, or bf16 - which is
    what conv2d() hands over inside a bf16 autocast region: bf16 storage, the bf16 matrix instructions in forward / input gradient,
    fp32 accumulation of the weight gradient.  The weight rows of both directions are made from the master weight by one launch
    (conv_weight_rows), the gradients of weight and bias are returned in fp32: no cast nodes in the autograd graph."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias, stride, pad):
        _need_cuda(x, weight, bias)
        if weight.dtype != torch.float32 or x.dtype not in (torch.float32, torch.bfloat16):
            raise CobevtHipError("training conv2d: fp32 master weight, x fp32 or bf16 (got %s, %s)" % (weight.dtype, x.dtype))
        xl = x.permute(0, 2, 3, 1).contiguous()
        n, h, w, _ = xl.shape
        cout, cin, kh, kw = weight.shape
        ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        strips = (USE_TRAIN_STRIPS and xl.dtype == torch.bfloat16 and kh == 3 and kw == 3 and pad == 1 and stride in (1, 2) and cin % 64 == 0
                  and cout % 8 == 0 and xl.numel() < 2 ** 31)
        # the input gradient of a stride-1 convolution is a 3x3 / pad-1 convolution again (Cout -> Cin channels)
        strips_d = USE_TRAIN_STRIPS and xl.dtype == torch.bfloat16 and kh == 3 and kw == 3 and pad == 1 and stride == 1 and cout % 64 == 0 \
            and cin % 8 == 0 and n * h * w * cout < 2 ** 31
        need_d = ctx.needs_input_grad[0]
        var_d = conv3_strips_plan(n, h, w, cout, cin, 1) if (strips_d and need_d) else -1
        rows = (USE_TRAIN_ROWS_GEMM and xl.dtype == torch.bfloat16 and kh == 1 and kw == 1 and stride == 1 and pad == 0 and cin % 8 == 0
                and cout % 8 == 0 and cin <= 512 and cout <= 512)
        if rows:
            # a dense projection: the inference row GEMM both ways (weights resident per workgroup, rows streamed once)
            frag_f, rows_d = linear_weight_frags(weight.reshape(cout, cin), True, need_d)
            out = _rows_gemm(xl.reshape(-1, cin), frag_f, cout, cin, bias).reshape(n, ho, wo, cout)
            var_d = -2 if need_d else -1
        elif strips:
            var_f = conv3_strips_plan(n, ho, wo, cin, cout, stride)
            op_f, op_d = conv3_weight_operand_pair(weight, var_f, var_d)      # both directions' operands from one launch
            out = _conv3_strips(xl, op_f, var_f, cout, bias, stride)
            rows_d = op_d if var_d >= 0 else conv_weight_rows(weight, xl.dtype, False, need_d)[1]
        else:
            rows_f, rows_d = conv_weight_rows(weight, xl.dtype, True, need_d and var_d < 0)
            out = _igemm_rows(xl, rows_f, cout, cin, kh, kw, bias, stride, pad, ho, wo)
            if var_d >= 0:
                rows_d = conv3_weight_operand(weight, var_d, True)
        ctx.save_for_backward(xl, rows_d)
        ctx.cfg = (stride, pad, bias is not None, tuple(weight.shape))
        ctx.var_d = var_d
        ctx.bias_dtype = None if bias is None else bias.dtype
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        xl, rows_d = ctx.saved_tensors
        stride, pad, has_bias, (cout, cin, kh, kw) = ctx.cfg
        n, h, w, _ = xl.shape
        dyl = dy.to(xl.dtype).permute(0, 2, 3, 1).contiguous()
        ho, wo = dyl.shape[1:3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            g = dyl
            if stride > 1:                       # zero-stuffed gradient map: dgrad of a strided conv = stride-1 conv on it
                g = torch.zeros((n, (ho - 1) * stride + 1, (wo - 1) * stride + 1, cout), device=dyl.device, dtype=dyl.dtype)
                g[:, ::stride, ::stride] = dyl
            if ctx.var_d == -2:
                dx = _rows_gemm(g.reshape(-1, cout), rows_d, cin, cout, None).reshape(n, h, w, cin).permute(0, 3, 1, 2)
            elif ctx.var_d >= 0:
                dx = _conv3_strips(g, rows_d, ctx.var_d, cin, None, 1).permute(0, 3, 1, 2)
            else:
                dx = _igemm_rows(g, rows_d, cin, cout, kh, kw, None, 1, kh - 1 - pad, h, w).permute(0, 3, 1, 2)
        chunks3 = -1
        if ctx.needs_input_grad[1] and USE_WGRAD3 and xl.dtype == torch.bfloat16 and kh == 3 and kw == 3 and stride == 1 and pad == 1:
            chunks3 = _L.load("").cobevt_conv_wgrad3_chunks(_ints([n, h, w, cin, cout]))
        chunks1 = -1
        if ctx.needs_input_grad[1] and USE_WGRAD3 and xl.dtype == torch.bfloat16 and kh == 1 and kw == 1 and stride == 1 and pad == 0:
            chunks1 = _L.load("").cobevt_linear_wgrad_chunks((ctypes.c_long * 3)(n * h * w, cin, cout))
        if chunks1 > 0:
            # a dense projection: dy^T x over the rows (csrc/wgrad3.hip), dw written, partial sums in a scratch buffer
            dw = torch.empty((cout, cin, 1, 1), device=dyl.device, dtype=torch.float32)
            scratch = torch.empty((chunks1, cout * cin), device=dyl.device, dtype=torch.float32)
            rc = _L.load("").cobevt_linear_wgrad(_p(xl), _p(dyl), _p(dw), _p(scratch), (ctypes.c_long * 4)(n * h * w, cin, cout, chunks1), _stream())
            _L.check(rc, "cobevt_linear_wgrad")
        elif chunks3 > 0:
            # straight from the channels-last maps (csrc/wgrad3.hip): dw is written, partial sums in a scratch buffer
            dw = torch.empty((cout, cin, kh, kw), device=dyl.device, dtype=torch.float32)
            scratch = torch.empty((chunks3, cout * cin * 9), device=dyl.device, dtype=torch.float32)
            rc = _L.load("").cobevt_conv_wgrad3(_p(xl), _p(dyl), _p(dw), _p(scratch), _ints([n, h, w, cin, cout, chunks3]), _stream())
            _L.check(rc, "cobevt_conv_wgrad3")
        elif ctx.needs_input_grad[1]:
            dw = _zeros((cout, cin, kh, kw), dyl.device, torch.float32)
            mode = wgrad_blocked_mode(kh, stride, pad, cin) if (USE_WGRAD_BLOCKED and xl.dtype == torch.bfloat16 and kh == kw) else None
            if mode is not None and not USE_WGRAD_BLOCKED_STRIDED and mode != 0:
                mode = None
            if mode is not None:
                x_blk, dy_blk, hp, nxb, ndb = blocked_operands(xl, dyl, kh, pad, stride, mode)
                dims = _ints([n, hp, nxb, cin, ho, ndb, cout, kh, stride, mode])
                rc = _L.load("").cobevt_conv_wgrad_blocked(_p(x_blk), _p(dy_blk), _p(dw), dims, _stream())
                _L.check(rc, "cobevt_conv_wgrad_blocked")
            else:
                dims = _ints([n, h, w, cin, ho, wo, cout, kh, stride, pad, ops.BF16 if xl.dtype == torch.bfloat16 else ops.FP32])
                rc = _L.load("").cobevt_conv_wgrad(_p(xl), _p(dyl), _p(dw), dims, _stream())
                _L.check(rc, "cobevt_conv_wgrad")
        if has_bias and ctx.needs_input_grad[2]:
            db = column_sum(dyl.reshape(-1, cout)).to(ctx.bias_dtype)
        return dx, dw, db, None, None


def conv2d(x, conv):
    """x through the nn.Conv2d container `conv` (square kernel / stride / padding, groups 1, dilation 1)"""
    if conv.groups != 1 or conv.dilation != (1, 1) or conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1] \
            or conv.padding[0] != conv.padding[1]:
        raise CobevtHipError("training conv2d: square kernel / stride / padding, groups 1, dilation 1 only")
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        # a bf16 autocast region: the convolution computes in bf16 like torch's own conv would, from the fp32 master weights (the
        # parameter receives an fp32 gradient); fp16 regions stay on the fp32 kernels (_amp_fwd)
        with torch.autocast("cuda", enabled=False):
            return Conv2dFn.apply(x.to(torch.bfloat16), _master(conv.weight), conv.bias, conv.stride[0], conv.padding[0])
    return Conv2dFn.apply(x, _master(conv.weight), conv.bias, conv.stride[0], conv.padding[0])


# ----------------------------------------------------------------------------------------------
# glue between the convolutions: BatchNorm (+ residual + ReLU), max-pool, PixelUnshuffle, nearest up-sampling, the STTF warp
# (csrc/train_glue.hip).  Tensors are (N, C, H, W)-shaped in channels-last memory, fp32 or bf16 (bf16 autocast regions).
# ----------------------------------------------------------------------------------------------
USE_TORCH_GLUE = False       # True: BatchNorm / max-pool / PixelUnshuffle / up-sampling through torch's ops (A/B and diagnostics only)
TORCH_GLUE_BN_IDS = set()    # ... or for the BatchNorm containers with these id()s (tools/train_grad_diag.py bisects by module name)
TORCH_GLUE_OPS = set()       # ... or only some of them: {"bn", "bn_res", "bn_plain", "pool", "shuffle", "up", "sttf"} (tools/train_grad_diag.py)


def _nhwc(x):
    """(N, C, H, W)-shaped -> contiguous (N, H, W, C) (no copy for channels-last memory); fp16 is widened to fp32"""
    if x.dtype == torch.float16:
        x = x.float()
    return x.permute(0, 2, 3, 1).contiguous()


class BatchNormActFn(torch.autograd.Function):
    """act(BatchNorm2d(x) [+ residual]) with nn.BatchNorm2d semantics: batch statistics and the in-place running-stat update when
    `training`, the frozen running statistics otherwise; act 1 = ReLU.  Forward: cobevt_channel_sums -> cobevt_bn_finalize ->
    cobevt_bn_apply; backward: cobevt_bn_backward (per-channel reductions in fp64, then dx / d(residual)) - what torch's batch_norm
    + add + relu do under train_camera.py:143-179 for torchvision's BasicBlock / Bottleneck and the NaiveDecoder."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, bn, training, act):
        _need_cuda(x, residual, gamma, beta)
        xl = _nhwc(x)
        rl = None if residual is None else _nhwc(residual.to(xl.dtype))
        n, h, w, c = xl.shape
        rows = n * h * w
        dev = xl.device
        lib = _L.load("")
        dt = ops.dcode(xl.dtype)
        scale, shift = torch.empty(c, device=dev), torch.empty(c, device=dev)
        mean, rstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
        g = None if gamma is None else _f32c(gamma.float(), "gamma")
        b = None if beta is None else _f32c(beta.float(), "beta")
        track = bn.track_running_stats and bn.running_mean is not None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        if training:
            # batch statistics: sums of x - running_mean (when there is one: no cancellation in the variance, fp32 partial sums suffice)
            # per workgroup, then one launch that reduces them and finishes every channel incl. the running-stat update
            _L.check(lib.cobevt_bn_batch_stats(_p(xl), _p(g), _p(b), _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                                               _p(scale), _p(shift), _p(mean), _p(rstd), _p(_scratch(c, dev)), _SCRATCH_BLOCKS, dt, rows, c,
                                               ctypes.c_float(bn.eps), ctypes.c_float(momentum),
                                               _p(bn.num_batches_tracked) if (track and bn.num_batches_tracked is not None) else None, _stream()),
                     "cobevt_bn_batch_stats")
            if track:
                # the kernel wrote the running statistics through raw pointers: tell torch, so that everything keyed on a buffer's
                # version counter - HipModule._plan's folded eval-mode weights, the captured-graph fingerprint of
                # host.pipeline.AgentCountPlans - sees the update even when no optimizer step touches the sibling parameters
                # (a frozen backbone with BatchNorm in train mode)
                for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
                    if t is not None:
                        torch.autograd.graph.increment_version(t)
        else:
            _L.check(lib.cobevt_bn_finalize(None, None, _p(g), _p(b), _p(bn.running_mean), _p(bn.running_var), _p(scale), _p(shift), _p(mean),
                                            _p(rstd), c, rows, ctypes.c_float(bn.eps), ctypes.c_float(momentum), 0, 0, None, _stream()),
                     "cobevt_bn_finalize")
        y = torch.empty_like(xl)
        _L.check(lib.cobevt_bn_apply(_p(xl), _p(rl), _p(scale), _p(shift), _p(y), dt, rows, c, int(act), _stream()), "cobevt_bn_apply")
        ctx.save_for_backward(xl, y if act else None, mean, rstd, g)
        ctx.cfg = (int(act), int(training), residual is not None, gamma is not None, beta is not None,
                   None if gamma is None else gamma.dtype)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xl, y, mean, rstd, g = ctx.saved_tensors
        act, training, has_res, has_g, has_b, pdt = ctx.cfg
        n, h, w, c = xl.shape
        rows = n * h * w
        dyl = _nhwc(dy.to(xl.dtype))
        acc = torch.empty((2, c), device=xl.device, dtype=torch.float64)
        f = torch.empty((2, c), device=xl.device, dtype=torch.float32) if (has_g or has_b) else None
        dx = torch.empty_like(xl)
        dres = torch.empty_like(xl) if has_res else None
        lib = _L.load("")
        _L.check(lib.cobevt_bn_backward(_p(xl), _p(y), _p(dyl), _p(mean), _p(rstd), _p(g), _p(acc), _p(f), _p(_scratch(c, xl.device)),
                                        _SCRATCH_BLOCKS, _p(dx), _p(dres), ops.dcode(xl.dtype), rows, c, act, training, _stream()),
                 "cobevt_bn_backward")
        dg = f[0].to(pdt) if has_g else None
        db = f[1].to(pdt) if has_b else None
        return (dx.permute(0, 3, 1, 2), None if dres is None else dres.permute(0, 3, 1, 2), dg, db, None, None, None)


def batch_norm_act(x, bn, residual=None, relu=False):
    """relu?(bn(x) [+ residual]) through the nn.BatchNorm2d container `bn` (its own .training flag decides the statistics)"""
    cumulative = bn.momentum is None and bn.training and bn.track_running_stats   # running stats as a cumulative average (1 / n)
    diagnostic = (USE_TORCH_GLUE or id(bn) in TORCH_GLUE_BN_IDS or "bn" in TORCH_GLUE_OPS or ("bn_res" in TORCH_GLUE_OPS and residual is not None)
                  or ("bn_plain" in TORCH_GLUE_OPS and residual is None))
    if (x.shape[1] % 8 or cumulative) and not diagnostic:
        # no silent library path behind a shape: the BatchNorm kernels walk 8-channel groups and keep an
        # exponential running average; neither case occurs in any shipped config
        raise CobevtHipError("BatchNorm2d(%d%s): the HIP training kernels need a channel count that is a multiple of 8 and a numeric "
                             "momentum; set cobevt_amd.autograd.USE_TORCH_GLUE = True to run this layer through torch's own ops"
                             % (x.shape[1], ", momentum=None" if cumulative else ""))
    if diagnostic:
        # a diagnostic switch (tools/train_grad_diag.py): torch's ops
        F = torch.nn.functional
        y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
        y = y + residual if residual is not None else y
        return F.relu(y) if relu else y
    return BatchNormActFn.apply(x, residual, bn.weight, bn.bias, bn, bool(bn.training or not bn.track_running_stats), 1 if relu else 0)


class GroupMeanFn(torch.autograd.Function):
    """mean over dim 1 of a contiguous (B, n, ...) tensor (fp32 or bf16, fp32 arithmetic): the camera mean of CrossWinAttention
    (fax_modules.py:243) - cobevt_group_mean in both directions (torch: a reduce kernel, and div + expand + copy backward)"""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x if x.is_contiguous() else x.contiguous()
        b, n = x.shape[:2]
        inner = x.numel() // (b * n)
        out = torch.empty((b,) + tuple(x.shape[2:]), device=x.device, dtype=x.dtype)
        _L.check(_L.load("").cobevt_group_mean(_p(x), _p(out), ops.dcode(x.dtype), b, n, inner, 0, _stream()), "cobevt_group_mean")
        ctx.shape = tuple(x.shape)
        ctx.dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dy):
        b, n = ctx.shape[:2]
        dy = dy.to(ctx.dtype)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty(ctx.shape, device=dy.device, dtype=dy.dtype)
        _L.check(_L.load("").cobevt_group_mean(_p(dy), _p(dx), ops.dcode(dy.dtype), b, n, dy.numel() // b, 1, _stream()), "cobevt_group_mean")
        return dx


def group_mean(x):
    """x.mean(dim=1) for (B, n, ...) tensors with 8 | the inner size, fp32 / bf16 (anything else: torch)"""
    if x.dim() >= 2 and x.shape[1] == 1:
        return x.squeeze(1)                             # one slab: a view in both directions
    if x.dim() < 3 or x.dtype not in (torch.float32, torch.bfloat16) or (x.numel() // (x.shape[0] * x.shape[1])) % 8:
        return x.mean(dim=1)
    with torch.autocast("cuda", enabled=False):
        return GroupMeanFn.apply(x)


USE_FAX_BEV_QUERY = True      # the BEV query of CrossViewSwapAttention as one kernel per direction


class FaxBevQueryFn(torch.autograd.Function):
    """The geometry embeddings of CrossViewSwapAttention (fax_modules.py:330-372) as one kernel per direction (csrc/train_fax.hip):
    out[b, cam] = normalize(conv1x1(grid) - c[b, cam]) (+ x[b]), channels-last.  The BEV query: x (B, H, W, 128) fp32, grid (2, H, W) shared
    by the batch; the image (key) embedding: x None, grid (B, 4, h, w) the homogeneous ray directions per camera, n = 1.  weight
    (128, K, 1, 1) / bias (128,) | None the 1x1 convolution (fp32 masters), c (B * n, 128) the camera-centre embedding ->
    (B, n, H, W, 128) fp32.  round_bf16: inside a bf16 autocast region (the convolution and the subtraction round to bf16 there)."""

    @staticmethod
    def forward(ctx, x, grid, weight, bias, c, n, round_bf16):
        _need_cuda(x, grid, weight, bias, c)
        grid, c = _f32c(grid, "grid"), _f32c(c, "c")
        x = None if x is None else _f32c(x, "x")
        kd = weight.shape[1]
        w = _f32c(weight.detach().reshape(weight.shape[0], kd), "embedding weight")
        bb = None if bias is None else _f32c(bias.detach(), "embedding bias")
        per_batch = grid.dim() == 4
        H, W = grid.shape[-2:]
        B = grid.shape[0] if per_batch else x.shape[0]
        d = w.shape[0]
        out = torch.empty((B, n, H, W, d), device=grid.device, dtype=torch.float32)
        dims = _ints([B, n, H, W, d, int(round_bf16), kd, int(per_batch)])
        _L.check(_L.load("").cobevt_fax_bev_query_train(_p(grid), _p(w), _p(bb), _p(c), _p(x), _p(out), dims, _stream()), "cobevt_fax_bev_query_train")
        ctx.save_for_backward(grid, w, bb, c)
        ctx.cfg = (B, n, H, W, d, int(round_bf16), kd, int(per_batch), tuple(weight.shape), bias is not None, x is not None)
        return out

    @staticmethod
    def backward(ctx, dq):
        grid, w, bb, c = ctx.saved_tensors
        B, n, H, W, d, rb, kd, per_batch, wshape, has_bias, has_x = ctx.cfg
        dq = _f32c(dq.float(), "dq")
        dx = torch.empty((B, H, W, d), device=dq.device, dtype=torch.float32) if has_x else None
        dw = _zeros((d, kd), dq.device, torch.float32)
        db = _zeros(d, dq.device, torch.float32) if has_bias else None
        dc = _zeros((B * n, d), dq.device, torch.float32)
        _L.check(_L.load("").cobevt_fax_bev_query_train_bwd(_p(grid), _p(w), _p(bb), _p(c), _p(dq), _p(dx), _p(dw), _p(db), _p(dc),
                                                          _ints([B, n, H, W, d, rb, kd, per_batch]), _stream()), "cobevt_fax_bev_query_train_bwd")
        return dx, None, dw.reshape(wshape), db, dc, None, None


def fax_bev_query_fusable(x_l, conv, n):
    return (x_l.dim() == 4 and x_l.shape[-1] == 128 and 1 <= n <= 8 and tuple(conv.weight.shape[1:]) == (2, 1, 1) and conv.weight.shape[0] == 128
            and conv.weight.dtype == torch.float32)


def fax_bev_query(x_l, grid2, conv, c_embed, n):
    """x_l (B, H, W, 128) channels-last, grid2 (2, H, W), conv = the nn.Conv2d(2, 128, 1) bev_embed container, c_embed (B * n, 128) ->
    (B, n, H, W, 128) fp32 (FaxBevQueryFn)"""
    mode = _autocast_mode()
    with torch.autocast("cuda", enabled=False):
        return FaxBevQueryFn.apply(x_l.float(), grid2.float(), conv.weight, conv.bias, c_embed.float(), int(n), mode == "bf16")


def fax_img_embed_fusable(dd, conv):
    return (dd.dim() == 4 and dd.shape[1] == 4 and tuple(conv.weight.shape) == (128, 4, 1, 1) and conv.weight.dtype == torch.float32)


def fax_img_embed(dd, conv, c_embed):
    """dd (BN, 4, h, w) homogeneous ray directions (no gradient), conv = the nn.Conv2d(4, 128, 1, bias=False) img_embed container,
    c_embed (BN, 128) -> normalize(conv(dd) - c_embed) as (BN, h, w, 128) fp32 channels-last (FaxBevQueryFn with one "camera" per batch element)"""
    mode = _autocast_mode()
    with torch.autocast("cuda", enabled=False):
        out = FaxBevQueryFn.apply(None, dd.detach().float().contiguous(), conv.weight, conv.bias, c_embed.float(), 1, mode == "bf16")
    return out.reshape(dd.shape[0], dd.shape[2], dd.shape[3], out.shape[-1])


class MaxPool3x3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) (the ResNet stem, resnet_ms.py:70): cobevt_maxpool3x3s2 / cobevt_maxpool3x3s2_bwd_t"""

    @staticmethod
    def forward(ctx, x):
        xl = _nhwc(x)
        ctx.save_for_backward(xl)
        return ops.maxpool3x3s2(xl).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        (xl,) = ctx.saved_tensors
        n, h, w, c = xl.shape
        dyl = _nhwc(dy.to(xl.dtype))
        dx = torch.empty_like(xl)
        _L.check(_L.load("").cobevt_maxpool3x3s2_bwd_t(_p(xl), _p(dyl), _p(dx), ops.dcode(xl.dtype), n, h, w, c, _stream()),
                 "cobevt_maxpool3x3s2_bwd_t")
        return dx.permute(0, 3, 1, 2)


def max_pool3x3s2(x):
    if USE_TORCH_GLUE or "pool" in TORCH_GLUE_OPS:
        return torch.nn.functional.max_pool2d(x, 3, 2, 1)
    return MaxPool3x3s2Fn.apply(x)


def _pixel_unshuffle(tl, inverse):
    n, h, w, c = tl.shape
    if inverse:
        ho, wo, cc = h, w, c // 4
        out = torch.empty((n, 2 * h, 2 * w, cc), device=tl.device, dtype=tl.dtype)
    else:
        ho, wo, cc = h // 2, w // 2, c
        out = torch.empty((n, ho, wo, 4 * c), device=tl.device, dtype=tl.dtype)
    _L.check(_L.load("").cobevt_pixel_unshuffle2_nhwc(_p(tl), _p(out), ops.dcode(tl.dtype), n, ho, wo, cc, int(inverse), _stream()),
             "cobevt_pixel_unshuffle2_nhwc")
    return out


class PixelUnshuffle2Fn(torch.autograd.Function):
    """nn.PixelUnshuffle(2) (fax_modules.py:479): a permutation, its own inverse in backward"""

    @staticmethod
    def forward(ctx, x):
        xl = _nhwc(x)
        if xl.shape[1] % 2 or xl.shape[2] % 2:
            raise CobevtHipError("PixelUnshuffle(2) needs even map sizes")
        return _pixel_unshuffle(xl, False).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        return _pixel_unshuffle(_nhwc(dy), True).permute(0, 3, 1, 2)


def pixel_unshuffle2(x):
    if USE_TORCH_GLUE or "shuffle" in TORCH_GLUE_OPS:
        return torch.nn.functional.pixel_unshuffle(x, 2)
    return PixelUnshuffle2Fn.apply(x)


def _upsample2(tl, backward):
    n, h, w, c = tl.shape
    if backward:
        h, w = h // 2, w // 2
        out = torch.empty((n, h, w, c), device=tl.device, dtype=tl.dtype)
    else:
        out = torch.empty((n, 2 * h, 2 * w, c), device=tl.device, dtype=tl.dtype)
    _L.check(_L.load("").cobevt_upsample_nearest2_nhwc(_p(tl), _p(out), ops.dcode(tl.dtype), n, h, w, c, int(backward), _stream()),
             "cobevt_upsample_nearest2_nhwc")
    return out


class UpsampleNearest2Fn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode='nearest') (naive_decoder.py:84); backward = the 2 x 2 block sums"""

    @staticmethod
    def forward(ctx, x):
        return _upsample2(_nhwc(x), False).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        return _upsample2(_nhwc(dy), True).permute(0, 3, 1, 2)


def upsample_nearest2(x):
    if USE_TORCH_GLUE or "up" in TORCH_GLUE_OPS:
        return torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    return UpsampleNearest2Fn.apply(x)


class SttfWarpFn(torch.autograd.Function):
    """STTF warp (corpbevt.py:28-64) of per-agent maps into the ego frames: the inference kernel cobevt_sttf_warp forward, its adjoint
    cobevt_sttf_warp_bwd backward.  With record_len (int32 device (B,)): f is the un-grouped agent batch (agents, H, W, C) and
    regroup (fuse_utils.py:8-61) is folded in -> (B, max_cav, H, W, C); without: f is (B, L, H, W, C).  No gradient flows into the
    poses (train_camera.py feeds them as data)."""

    @staticmethod
    def forward(ctx, f, tm, record_len, max_cav, discrete_ratio, downsample_rate):
        fl = _f32c(f.float(), "features")
        tm = _f32c(tm.float(), "transformation_matrix")
        if record_len is not None:
            out, _, _ = ops.sttf_warp(fl, tm, None, discrete_ratio, downsample_rate, want_mask=False, record_len=record_len,
                                      max_cav=max_cav)
            b, l = record_len.shape[0], int(max_cav)
        else:
            out, _ = ops.sttf_warp(fl, tm, None, discrete_ratio, downsample_rate, want_mask=False)
            b, l = fl.shape[:2]
        ctx.save_for_backward(tm, record_len)
        ctx.cfg = (tuple(fl.shape), b, l, float(discrete_ratio), float(downsample_rate), f.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        tm, record_len = ctx.saved_tensors
        shape, b, l, ratio, rate, dt = ctx.cfg
        h, w, c = shape[-3:]
        dout = _f32c(dout.float(), "dout")
        dx = torch.zeros(shape, device=dout.device, dtype=torch.float32)
        _L.check(_L.load("").cobevt_sttf_warp_bwd(_p(dout), _p(tm), _p(record_len), _p(dx), b, l, h, w, c, ctypes.c_float(ratio),
                                                ctypes.c_float(rate), _stream()), "cobevt_sttf_warp_bwd")
        return dx.to(dt), None, None, None, None, None


def sttf_warp(f, tm, record_len, max_cav, discrete_ratio, downsample_rate):
    return SttfWarpFn.apply(f, tm, record_len, max_cav, discrete_ratio, downsample_rate)


class WeightedCrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(weight=w)(logits (N, C, H, W) fp32, target (N, H, W)) with the HIP forward (cobevt_weighted_cross_entropy)
    and backward (cobevt_weighted_cross_entropy_bwd): vanilla_seg_loss.py:18-23,58-70 under train_camera.py:166-173."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, logits, target, weight):
        if logits.dtype != torch.float32:
            raise CobevtHipError("the training slice is fp32")
        loss, stats, x, y, wt = ops.weighted_cross_entropy(logits, target, weight, want_stats=True)
        ctx.save_for_backward(x, y, wt, stats)
        return loss.clone()

    @staticmethod
    @_amp_bwd
    def backward(ctx, dloss):
        x, y, wt, stats = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        up = dloss.reshape(1).to(torch.float32).contiguous()
        rc = _L.load("").cobevt_weighted_cross_entropy_bwd(_p(x), _p(y), _p(wt), _p(stats), _p(up), _p(dx), n, c, h * w, _stream())
        _L.check(rc, "cobevt_weighted_cross_entropy_bwd")
        return dx, None, None


def weighted_cross_entropy(logits, target, weight):
    # explicit (differentiable) cast: the criterion is commonly called OUTSIDE the autocast region, where custom_fwd's
    # cast_inputs no longer applies and a bf16 / fp16 head output would reach the fp32 kernel as it is
    return WeightedCrossEntropyFn.apply(logits.float(), target, weight)


# ----------------------------------------------------------------------------------------------
# nuScenes SinBEVT pieces (csrc/train_nusc.hip): swish, depthwise convolution, align_corners bilinear resize, sigmoid focal loss
# ----------------------------------------------------------------------------------------------
class SwishFn(torch.autograd.Function):
    """x sigmoid(x) (efficientnet-pytorch's swish) on any contiguous tensor whose element count is a multiple of 8: cobevt_swish both ways"""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        xc = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
        # elementwise over memory: any dense layout will do (row-major, or the (N, C, H, W)-shaped channels-last views of this path)
        if not (xc.is_contiguous() or (xc.dim() == 4 and xc.is_contiguous(memory_format=torch.channels_last))):
            xc = xc.contiguous()
        out = torch.empty_like(xc)                          # same strides
        _L.check(_L.load("").cobevt_swish(_p(xc), None, _p(out), ops.dcode(xc.dtype), xc.numel(), _stream()), "cobevt_swish")
        ctx.save_for_backward(xc)
        return out

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        g = dy.to(xc.dtype)
        if g.stride() != xc.stride():                       # bring the gradient into x's memory order
            g = torch.empty_like(xc).copy_(g)
        dx = torch.empty_like(xc)
        _L.check(_L.load("").cobevt_swish(_p(xc), _p(g), _p(dx), ops.dcode(xc.dtype), xc.numel(), _stream()), "cobevt_swish")
        return dx


def swish(x):
    if x.numel() % 8:
        return x * torch.sigmoid(x)
    return SwishFn.apply(x)


class DepthwiseConvFn(torch.autograd.Function):
    """k x k depthwise convolution (groups == channels, TensorFlow-"same" static padding: `pad` = (before, after) zero rows / columns on
    each axis) on (N, C, H, W)-shaped tensors in channels-last memory, fp32 or bf16: cobevt_depthwise_conv_nhwc forward; the input
    gradient is the same kernel on the flipped taps (on the zero-stuffed gradient when strided), the weight gradient cobevt_depthwise_wgrad.
    weight: the fp32 (C, 1, k, k) parameter."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, stride, pad):
        _need_cuda(x, weight)
        xl = _nhwc(x)
        n, h, w, c = xl.shape
        k = weight.shape[-1]
        if c % 8 or weight.shape[0] != c or weight.shape[1] != 1 or weight.shape[2] != k or k not in (3, 5):
            raise CobevtHipError("training depthwise conv: (C, 1, k, k) weight with k = 3 / 5 and a multiple of 8 channels")
        ho, wo = (h + pad[0] + pad[1] - k) // stride + 1, (w + pad[0] + pad[1] - k) // stride + 1
        taps = weight.detach().float()[:, 0].permute(1, 2, 0).reshape(k * k, c).contiguous()          # [tap][C]
        zero = _zeros(c, xl.device, torch.float32)
        out = torch.empty((n, ho, wo, c), device=xl.device, dtype=xl.dtype)
        dims = _ints([ops.dcode(xl.dtype), n, h, w, c, k, stride, pad[0], pad[0], ho, wo, 0])
        _L.check(_L.load("").cobevt_depthwise_conv_nhwc(_p(xl), _p(taps), _p(zero), _p(out), dims, _stream()), "cobevt_depthwise_conv_nhwc")
        ctx.save_for_backward(xl, taps, zero)
        ctx.cfg = (int(stride), (int(pad[0]), int(pad[1])), k, weight.dtype)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @_amp_bwd
    def backward(ctx, dy):
        xl, taps, zero = ctx.saved_tensors
        stride, pad, k, wdt = ctx.cfg
        n, h, w, c = xl.shape
        dyl = _nhwc(dy.to(xl.dtype))
        ho, wo = dyl.shape[1:3]
        lib = _L.load("")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            g = dyl
            gh, gw = h + pad[0] + pad[1] - k + 1, w + pad[0] + pad[1] - k + 1          # the stride-1 output extent of the forward
            if stride > 1 or (gh, gw) != (ho, wo):           # zero-stuffed gradient map on that extent
                g = torch.zeros((n, gh, gw, c), device=dyl.device, dtype=dyl.dtype)
                g[:, :(ho - 1) * stride + 1:stride, :(wo - 1) * stride + 1:stride] = dyl
            flipped = taps.reshape(k, k, c).flip(0, 1).reshape(k * k, c).contiguous()
            dx = torch.empty_like(xl)
            # dx[i] = sum_a dy_stuffed[i + pad0 - a] w[a]  =  correlation of the stuffed gradient with the flipped taps, top / left pad k - 1 - pad0
            dims = _ints([ops.dcode(xl.dtype), n, g.shape[1], g.shape[2], c, k, 1, k - 1 - pad[0], k - 1 - pad[0], h, w, 0])
            _L.check(lib.cobevt_depthwise_conv_nhwc(_p(g), _p(flipped), _p(zero), _p(dx), dims, _stream()), "cobevt_depthwise_conv_nhwc")
            dx = dx.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dwt = _zeros((k * k, c), xl.device, torch.float32)
            dims = _ints([ops.dcode(xl.dtype), n, h, w, c, k, stride, pad[0], pad[0], ho, wo])
            _L.check(lib.cobevt_depthwise_wgrad(_p(xl), _p(dyl), _p(dwt), dims, _stream()), "cobevt_depthwise_wgrad")
            dw = dwt.reshape(k, k, c).permute(2, 0, 1)[:, None].contiguous().to(wdt)
        return dx, dw, None, None


def depthwise_conv2d(x, conv, pad):
    """x through the depthwise nn.Conv2d container `conv` (groups == channels, no bias) with static "same" padding pad = (before, after)"""
    if conv.groups != conv.in_channels or conv.bias is not None or conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1]:
        raise CobevtHipError("training depthwise conv: groups == channels, square kernel / stride, no bias")
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        with torch.autocast("cuda", enabled=False):
            return DepthwiseConvFn.apply(x.to(torch.bfloat16), _master(conv.weight), conv.stride[0], tuple(pad))
    return DepthwiseConvFn.apply(x, _master(conv.weight), conv.stride[0], tuple(pad))


class ResizeBilinearFn(torch.autograd.Function):
    """align_corners bilinear resize of an (N, C, H, W)-shaped tensor (nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
    nuscenes decoder.py:12): cobevt_resize_nhwc forward, its adjoint cobevt_resize_bilinear_bwd backward"""

    @staticmethod
    def forward(ctx, x, ho, wo):
        xl = _nhwc(x)
        ctx.shape = tuple(xl.shape)
        ctx.dt = xl.dtype
        return ops.resize_nhwc(xl, ho, wo, "bilinear").permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c = ctx.shape
        dyl = _nhwc(dy.to(ctx.dt))
        dx = torch.zeros((n, h, w, c), device=dyl.device, dtype=torch.float32)
        _L.check(_L.load("").cobevt_resize_bilinear_bwd(_p(dyl), _p(dx), ops.dcode(dyl.dtype), n, h, w, c, dyl.shape[1], dyl.shape[2], _stream()),
                 "cobevt_resize_bilinear_bwd")
        return dx.to(ctx.dt).permute(0, 3, 1, 2), None, None


def resize_bilinear(x, ho, wo):
    c = x.shape[1]
    g = c >> 3
    if c % 8 or g > 64 or (g & (g - 1)):
        if USE_TORCH_GLUE:
            return torch.nn.functional.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True)
        raise CobevtHipError("resize_bilinear: the HIP kernel takes 8 * 2^k <= 512 channels (got %d); set cobevt_amd.autograd."
                             "USE_TORCH_GLUE = True to run this resize through torch's interpolate" % c)
    return ResizeBilinearFn.apply(x, int(ho), int(wo))


class SigmoidFocalLossFn(torch.autograd.Function):
    """mean sigmoid focal loss over the visible pixels (nuscenes losses.py:27-84): cobevt_sigmoid_focal_loss / _bwd; pred (N, C, hw) fp32"""

    @staticmethod
    def forward(ctx, pred, label, vis, masks, cfg):
        min_visibility, alpha, gamma, soft = cfg
        _need_cuda(pred, label)
        n, c, hw = pred.shape
        p32 = _f32c(pred, "pred")
        nl = label.shape[1]
        scratch = torch.empty(2 * n * ((hw + 2047) // 2048), device=pred.device, dtype=torch.float32)
        out = torch.empty(3, device=pred.device, dtype=torch.float32)
        rc = _L.load("").cobevt_sigmoid_focal_loss(_p(p32), _p(label), _p(vis), _p(masks), _p(scratch), _p(out), n, c, nl, hw, min_visibility,
                                                 ctypes.c_float(alpha), ctypes.c_float(gamma), int(soft), _stream())
        _L.check(rc, "cobevt_sigmoid_focal_loss")
        ctx.save_for_backward(p32, label, vis, masks, out)
        ctx.cfg = cfg
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        p32, label, vis, masks, out = ctx.saved_tensors
        min_visibility, alpha, gamma, soft = ctx.cfg
        n, c, hw = p32.shape
        dp = torch.empty_like(p32)
        gs = g.reshape(1).float().contiguous()
        rc = _L.load("").cobevt_sigmoid_focal_loss_bwd(_p(p32), _p(label), _p(vis), _p(masks), _p(out), _p(gs), _p(dp), n, c, label.shape[1], hw,
                                                     min_visibility, ctypes.c_float(alpha), ctypes.c_float(gamma), int(soft), _stream())
        _L.check(rc, "cobevt_sigmoid_focal_loss_bwd")
        return dp, None, None, None, None


class PairwiseWarpFn(torch.autograd.Function):
    """ops.pairwise_warp (every agent's map in every other agent's frame, v2v_fuse.py:59-103) with its adjoint as backward: x (N, H, W, C)
    channels-last fp32 -> nb (B, L, L, H, W, C); the ROI mask carries no gradient (ops.pairwise_warp returns it)"""

    @staticmethod
    def forward(ctx, x, pairwise, record_len, max_cav, discrete_ratio, downsample_rate):
        nb, _ = ops.pairwise_warp(x, pairwise, record_len, max_cav, discrete_ratio, downsample_rate)
        ctx.save_for_backward(pairwise, record_len)
        ctx.cfg = (tuple(x.shape), int(max_cav), float(discrete_ratio), float(downsample_rate))
        return nb

    @staticmethod
    def backward(ctx, dnb):
        pairwise, record_len = ctx.saved_tensors
        (n, h, w, c), l, dr, ds = ctx.cfg
        dnb = dnb.contiguous()
        dx = torch.zeros((n, h, w, c), device=dnb.device, dtype=torch.float32)
        rc = _L.load("").cobevt_pairwise_warp_bwd(_p(dnb), _p(pairwise), _p(record_len), _p(dx), ops.dcode(dnb.dtype), record_len.shape[0], l,
                                                h, w, c, ctypes.c_float(dr), ctypes.c_float(ds), _stream())
        _L.check(rc, "cobevt_pairwise_warp_bwd")
        return dx.to(dnb.dtype), None, None, None, None, None


def conv2d_weight(x, weight, bias, stride=1, pad=0):
    """the training convolution on a weight TENSOR (a differentiable function of parameters: sliced / re-indexed taps) instead of an
    nn.Conv2d container"""
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        with torch.autocast("cuda", enabled=False):
            return Conv2dFn.apply(x.to(torch.bfloat16), _master(weight).contiguous(), bias, stride, pad)
    return Conv2dFn.apply(x, _master(weight).contiguous(), bias, stride, pad)


def _env_flags():
    """COBEVT_TRAIN_FLAGS="USE_TORCH_OPERAND_PREP=1,USE_WGRAD_BLOCKED=0": the module's USE_* switches from the environment (same-job A/B
    runs of tools/train_probe.py; never set in production)"""
    import os
    for item in os.environ.get("COBEVT_TRAIN_FLAGS", "").split(","):
        if not item.strip():
            continue
        k, _, v = item.partition("=")
        k = k.strip()
        if not k.startswith("USE_") or k not in globals() or not isinstance(globals()[k], bool):
            raise CobevtHipError("COBEVT_TRAIN_FLAGS: unknown switch %r" % k)
        globals()[k] = bool(int(v))


_env_flags()
