"""Training slice: torch.autograd.Functions over the HIP forward / backward kernels.

The reference trains through torch autograd (opv2v/opencood/tools/train_camera.py:143-179: model.train(), loss.backward(),
optimizer.step()).  Here the attention core (gathered window / dilated-grid partition, relative-position bias, key mask,
softmax, PV, partition reverse), LayerNorm and GELU run as HIP kernels in both directions (fp32 storage, exact-fp32 MFMA:
csrc/attention.hip + attention_bwd.hip, csrc/elementwise.hip + train_rows.hip); the dense projections are plain GEMMs and go
to the library (rocBLAS through torch.matmul) in both directions.  fp32 only: the bf16 inference layouts (fragment-ordered
weights, folded norms) are not differentiable containers.

No CPU path: every Function raises on CPU tensors (lib.CobevtHipError) like the inference ops do.
"""
import ctypes

import torch

from . import lib as _L
from . import ops
from .lib import CobevtHipError

_p, _stream, _ints, _need_cuda = ops._p, ops._stream, ops._ints, ops._need_cuda


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


class WindowAttentionFn(torch.autograd.Function):
    """out = softmax(scale q k^T + bias[rel(q, k)] + mask) v over the gathered windows (cobevt_window_attention_lse /
    cobevt_window_attention_bwd).  q (Rq, d), k, v (Rk, d) token matrices (views with a row stride are accepted), bias_table
    (rows, heads) | None, mask fp32 | None (no gradient).  cfg = (qmap, kmap, omap, batch, heads, scale, bias_L, out_rows)."""

    @staticmethod
    def forward(ctx, q, k, v, bias_table, mask, cfg):
        qmap, kmap, omap, batch, heads, scale, bias_L, out_rows = cfg
        _need_cuda(q, k, v, bias_table, mask)
        ldq, ldk, ldv = _rows_view(q, "q"), _rows_view(k, "k"), _rows_view(v, "v")
        d = heads * 32
        if q.shape[1] != d or k.shape[1] != d or v.shape[1] != d:
            raise CobevtHipError("window attention: token width must be heads * 32")
        table = None if bias_table is None else _f32c(bias_table, "bias_table")
        mk = None if mask is None else _f32c(mask, "mask")
        L = qmap[6] * qmap[7]
        nq = qmap[1] * qmap[4] * qmap[5]
        out = torch.empty((out_rows, d), device=q.device, dtype=torch.float32)
        lse = torch.empty((batch, L, heads, nq), device=q.device, dtype=torch.float32)
        dims = _attn_dims(batch, heads, ldq, ldk, ldv, d, table, bias_L, qmap, kmap, omap)
        rc = _L.load().cobevt_window_attention_lse(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(table), _p(mk), dims,
                                                   ctypes.c_float(scale), _stream())
        _L.check(rc, "cobevt_window_attention_lse")
        ctx.save_for_backward(q, k, v, out, lse, table, mk)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, table, mk = ctx.saved_tensors
        qmap, kmap, omap, batch, heads, scale, bias_L, _ = ctx.cfg
        d = heads * 32
        dout = _f32c(dout, "dout")
        # dq is accumulated with atomics by the key tiles of a window; dk / dv rows are written once each
        dq = torch.zeros((q.shape[0], d), device=q.device, dtype=torch.float32)
        dk = torch.zeros((k.shape[0], d), device=q.device, dtype=torch.float32)
        dv = torch.zeros((v.shape[0], d), device=q.device, dtype=torch.float32)
        dbias = None if table is None else torch.zeros_like(table)
        dims = _attn_dims(batch, heads, q.stride(0), k.stride(0), v.stride(0), d, table, bias_L, qmap, kmap, omap)
        # the gradient buffers are dense (ld = d) while q / k / v may be strided views: the kernel shares one ld per
        # operand between the tensor and its gradient, so strided operands are compacted first
        if q.stride(0) != d or k.stride(0) != d or v.stride(0) != d:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            dims = _attn_dims(batch, heads, d, d, d, d, table, bias_L, qmap, kmap, omap)
        rc = _L.load().cobevt_window_attention_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), _p(dq), _p(dk), _p(dv),
                                                   _p(dbias), _p(table), _p(mk), dims, ctypes.c_float(scale), _stream())
        _L.check(rc, "cobevt_window_attention_bwd")
        return dq, dk, dv, dbias, None, None


def window_attention(q, k, v, qmap, kmap, omap, batch, heads, scale, out_rows, bias_table=None, bias_L=1, mask=None):
    cfg = (tuple(qmap), tuple(kmap), tuple(omap), int(batch), int(heads), float(scale), int(bias_L), int(out_rows))
    return WindowAttentionFn.apply(q, k, v, bias_table, mask, cfg)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension: cobevt_layernorm forward, cobevt_layernorm_bwd backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _f32c(x, "layernorm input")
        g, b = _f32c(gamma, "gamma"), _f32c(beta, "beta")
        y = ops.layernorm(x, g, b, eps)
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.zeros(C, device=x.device, dtype=torch.float32)
        db = torch.zeros(C, device=x.device, dtype=torch.float32)
        rc = _L.load().cobevt_layernorm_bwd(_p(x), _p(dy), _p(g), _p(dx), _p(dg), _p(db), rows, C, ctypes.c_float(ctx.eps),
                                            _stream())
        _L.check(rc, "cobevt_layernorm_bwd")
        return dx, dg, db, None


def layernorm(x, ln):
    """x through the nn.LayerNorm container `ln`"""
    return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)


class GeluFn(torch.autograd.Function):
    """nn.GELU() (exact erf form): cobevt_gelu in both directions."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x, "gelu input")
        _need_cuda(x)
        y = torch.empty_like(x)
        rc = _L.load().cobevt_gelu(_p(x), None, _p(y), x.numel(), _stream())
        _L.check(rc, "cobevt_gelu")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        dx = torch.empty_like(x)
        rc = _L.load().cobevt_gelu(_p(x), _p(dy), _p(dx), x.numel(), _stream())
        _L.check(rc, "cobevt_gelu")
        return dx


def gelu(x):
    return GeluFn.apply(x)


def linear(x, lin):
    """nn.Linear container: a plain GEMM in both directions -> the library (rocBLAS via torch), the one place the guide allows it."""
    _need_cuda(x)
    return torch.nn.functional.linear(x, lin.weight, lin.bias)


def dropout(x, p):
    """nn.Dropout in train mode (elementwise mask; torch's generator so seeds behave like the reference's)."""
    return torch.nn.functional.dropout(x, p, True) if p > 0 else x
