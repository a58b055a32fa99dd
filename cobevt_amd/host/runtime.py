"""Shared plumbing of the host modules: compute-dtype selection, lowered-plan caching, layout helpers."""
import contextlib

import torch
import torch.nn as nn

from .. import ops
from ..lib import CobevtHipError

_COMPUTE_DTYPE = torch.bfloat16
_MATRIX_PATH = "native"
MATRIX_PATHS = ("native", "split_bf16", "split_bf16_enc_fp16")
from .. import lib as _lib  # noqa: E402


def set_compute_dtype(dtype, matrix_path=None):
    """torch.bfloat16 (perf mode: bf16 storage + bf16 MFMA, fp32 accumulate) or torch.float32 (parity modes, fp32 storage):
    matrix_path "native" = exact v_mfma_f32_32x32x2_f32 (default), "split_bf16" = every matrix product of the inference
    kernels as two v_mfma_f32_32x32x16_bf16 over (hi, lo) bf16 halves of both operands (all four cross terms, |x - hi - lo| <=
    2^-17 |x|: ~1e-5 end to end instead of 1e-6, at 4x the matrix rate; served by libcobevt_hip_f32s.so, cobevt_amd/build.py).
    The string "fp32_split" is shorthand for (torch.float32, "split_bf16").
    "fp32_fast" = (torch.float32, "split_bf16_enc_fp16") (round 6): as "fp32_split", except that the ResNet encoder's convolutions -
    80 % of a frame's flops - run on v_mfma_f32_32x32x16_f16 with fp16 OPERANDS out of fp32 storage (libcobevt_hip_f32h.so;
    csrc/common.hpp COBEVT_F32_SPLIT == 2): the folded weights as one fp16 term, the activations as one fp16 value where a wave owns
    two k-groups per tap (the packed form: one MFMA per two k-groups - the strip kernels' 128-cout tiles, the BasicBlocks) and as an
    fp16 (hi, lo) pair elsewhere (stem, 64-cout tiles, 1x1 shortcuts).  fp32 accumulation, fp32 storage, every residual added in
    fp32; ~3e-4 max-rel on the 5-agent frame (inside the north-star's 1e-3, not the 1e-5 of "fp32_split").  Precondition: encoder
    activations and folded weights within fp16 range (|v| <= 65504), as under the reference's own fp16 autocast (train_camera.py:157-160)."""
    global _COMPUTE_DTYPE, _MATRIX_PATH
    if isinstance(dtype, str):
        alias = {"bf16": (torch.bfloat16, "native"), "fp32": (torch.float32, "native"), "fp32_split": (torch.float32, "split_bf16"),
                 "fp32_fast": (torch.float32, "split_bf16_enc_fp16")}
        if dtype not in alias:
            raise CobevtHipError("compute mode must be one of %s" % sorted(alias))
        dtype, mp = alias[dtype]
        matrix_path = mp if matrix_path is None else matrix_path
    matrix_path = "native" if matrix_path is None else matrix_path
    if matrix_path not in MATRIX_PATHS:
        raise CobevtHipError("matrix_path must be one of %s" % (MATRIX_PATHS,))
    ops.dcode(dtype)
    if matrix_path != "native" and dtype != torch.float32:
        raise CobevtHipError("the split-bf16 matrix paths belong to fp32 storage (bf16 storage IS the bf16 matrix path)")
    _COMPUTE_DTYPE, _MATRIX_PATH = dtype, matrix_path
    _lib.set_variant("" if matrix_path == "native" else "f32s")
    _lib.set_encoder_variant("f32h" if matrix_path == "split_bf16_enc_fp16" else None)


def get_compute_dtype():
    return _COMPUTE_DTYPE


def get_matrix_path():
    return _MATRIX_PATH


def get_compute_mode():
    """"bf16" | "fp32" | "fp32_split" | "fp32_fast": the key captured graphs are cached under (plans = lowered weights depend on the dtype only)"""
    if _COMPUTE_DTYPE == torch.bfloat16:
        return "bf16"
    return {"native": "fp32", "split_bf16": "fp32_split", "split_bf16_enc_fp16": "fp32_fast"}[_MATRIX_PATH]


@contextlib.contextmanager
def compute_dtype(dtype, matrix_path=None):
    prev = (get_compute_dtype(), get_matrix_path())
    set_compute_dtype(dtype, matrix_path)
    try:
        yield
    finally:
        set_compute_dtype(*prev)


_STRUCTURE_EPOCH = 0


def bump_structure_epoch():
    """A module gained, lost or replaced a tensor the kernels read (not an in-place update: those are seen through version
    counters).  Caches that list a model's tensors once (host.pipeline.AgentCountPlans) rebuild their list when this moves."""
    global _STRUCTURE_EPOCH
    _STRUCTURE_EPOCH += 1


def structure_epoch():
    return _STRUCTURE_EPOCH


def _install_structure_hooks():
    """Every way a module can gain or REPLACE a tensor or a child moves the epoch (AgentCountPlans lists the model's
    tensors once, and a replaced Parameter object - `m.weight = nn.Parameter(..)`, register_buffer, pruning / parametrize, BN fusing -
    kept the old tensor in that list, so the fingerprint stayed equal and a captured graph kept replaying the old weights).
    torch's global registration hooks fire from Module.register_parameter / register_buffer / register_module AND from
    Module.__setattr__ when a Parameter, a registered buffer or a child module is assigned, on every nn.Module - the parameter
    containers (nn.Linear, nn.Conv2d, ...) inside the HIP modules included.  HipModule._apply covers conversions that swap Parameter
    objects (torch.__future__.set_overwrite_module_params_on_conversion).  In-place updates are seen through version counters and
    `.data` writes are documented under HipModule.invalidate_plans."""
    import torch.nn.modules.module as _m

    def bump(*_args):
        bump_structure_epoch()
        return None
    for reg in ("register_module_parameter_registration_hook", "register_module_buffer_registration_hook",
                "register_module_module_registration_hook"):
        getattr(_m, reg)(bump)


_install_structure_hooks()


class GraphOwner(object):
    """Base of every object that owns captured HIP graphs (the runners of host/pipeline.py, host/train_graph.py).  Destroying a
    graph while a replay of it is still running on the GPU takes the process down on ROCm 7.2 - an abort a few launches later, seen in
    2 of 8 runs of tests/test_pipeline_gpu.py where `enable_graphs(False)` dropped a plan right after its replay - so an owner waits for
    the device before its graph objects go (`__del__` runs before the attributes are released).  Runners die rarely (a re-capture
    after a weight update, an evicted plan, the end of a run): the wait is off every hot path."""

    def __del__(self):
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
        except Exception:      # interpreter shutdown: torch may be half gone, and then so is every stream
            pass


class HipModule(nn.Module):
    """nn.Module whose parameters are plain torch containers (reference-compatible state_dict keys) and whose
    forward is HIP kernels.  Lowered weights ("plans": folded BatchNorm, [Cout][K] layout, compute dtype) are
    cached per (name, dtype, device) and rebuilt when a source parameter changes."""

    def __init__(self):
        super().__init__()
        self._plan_cache = {}

    def _plan(self, name, tensors, builder):
        tensors = [t for t in tensors if t is not None]
        dt, dev = get_compute_dtype(), tensors[0].device
        key = (name, dt, dev)
        ver = tuple((t._version, t.data_ptr()) for t in tensors)
        ent = self._plan_cache.get(key)
        if ent is None or ent[0] != ver:
            ent = (ver, builder(dt, dev))
            self._plan_cache[key] = ent
        return ent[1]

    def invalidate_plans(self):
        """Drop every cached plan of this module and its children.  The cache key sees in-place updates of a parameter
        (`p.mul_()`, optimizer steps, load_state_dict) through its version counter, but NOT writes that go through `.data`
        (`p.data.copy_()`, EMA / weight-surgery code, some checkpoint loaders): call this after such writes."""
        for m in self.modules():
            if isinstance(m, HipModule):
                m._plan_cache.clear()
            gp = getattr(m, "graph_plans", None)      # captured graphs hold the addresses of the dropped buffers
            if gp is not None:
                gp.clear()

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_plans()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        bump_structure_epoch()        # .to() / .half() / .float(): tensors may have been replaced (always re-list them)
        return out

    def _require_inference(self, *tensors):
        if self.training:
            raise CobevtHipError("%s.forward is the fused inference path: call .eval() first.  In train() mode this module runs "
                                 "inside its model's differentiable graph (cobevt_amd/host/training.py; every registry model trains), "
                                 "not through this stand-alone forward" % type(self).__name__)
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise CobevtHipError("%s.forward needs ROCm device tensors; the HIP path has no CPU fallback"
                                     % type(self).__name__)


def module_tensors(*mods):
    out = []
    for m in mods:
        if m is None:
            continue
        out.extend(p for p in m.parameters(recurse=True))
        out.extend(b for b in m.buffers(recurse=True) if torch.is_floating_point(b))
    return out


def conv_plan(owner, name, conv, bn=None, pre_bn=None, act=0, upsample=False, store_mode=0, smallc=False):
    """Plan for an nn.Conv2d container (+ following eval BatchNorm, + preceding pre-activation BN->ReLU)."""
    def build(dt, dev):
        return ops.ConvPlan(conv.weight, conv.bias, bn=bn, pre_bn=pre_bn, pre_relu=pre_bn is not None,
                            stride=conv.stride[0], pad=conv.padding[0], act=act, upsample=upsample,
                            store_mode=store_mode, dtype=dt, device=dev, smallc=smallc)
    return owner._plan(name, module_tensors(conv, bn, pre_bn), build)


def linear_plan(owner, name, lin, act=0, ln=None):
    """Plan for an nn.Linear container; ln = the nn.LayerNorm container applied to its input (folded into the plan)."""
    def build(dt, dev):
        return ops.ConvPlan(lin.weight, lin.bias, act=act, dtype=dt, device=dev, ln=ln)
    return owner._plan(name, module_tensors(lin, ln), build)


def f32_param(owner, name, tensor, shape=None):
    """fp32 contiguous device copy of a small parameter/buffer (LayerNorm affine, embedding tables, 1x1 geometry convs)."""
    def build(dt, dev):
        t = tensor.detach().to(device=dev, dtype=torch.float32)
        if shape is not None:
            t = t.reshape(shape)
        return t.contiguous()
    return owner._plan("f32:" + name, [tensor], build)


def layernorm(owner, name, ln, x):
    g = f32_param(owner, name + ".w", ln.weight)
    b = f32_param(owner, name + ".b", ln.bias)
    return ops.layernorm(x, g, b, ln.eps)


def to_nhwc(x):
    """(N,C,H,W)-shaped tensor -> contiguous (N,H,W,C) in the compute dtype (zero-copy for channels-last views)."""
    return ops.to_nhwc(x, get_compute_dtype())


def nchw_view(x):
    """contiguous (N,H,W,C) -> (N,C,H,W)-shaped view (no copy)."""
    return x.permute(0, 3, 1, 2)


def as_compute(x):
    """Contiguous tensor in the compute dtype (API-boundary cast; internal tensors already are)."""
    dt = get_compute_dtype()
    if x.dtype != dt:
        x = x.to(dt)
    return x if x.is_contiguous() else x.contiguous()


def like_input(y, ref):
    """Public sub-module forwards return the caller's dtype (the reference is fp32 in / fp32 out)."""
    return y if y.dtype == ref.dtype else y.to(ref.dtype)
