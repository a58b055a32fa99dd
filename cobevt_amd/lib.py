"""ctypes binding of libcobevt_hip.so (the C ABI declared in include/cobevt_hip.h).

There is deliberately no fallback: if the shared library is missing or a symbol is absent this raises.
Import torch before loading so the HIP runtime already mapped by torch (same SONAME) is reused.
"""
import ctypes
import os

import torch  # noqa: F401  (maps libamdhip64 first; the kernels must share torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcobevt_hip.so")
# A/B measurements inside one GPU job (box-to-box variance is larger than most kernel changes): COBEVT_HIP_LIB points at
# another build of the same ABI, e.g. the previous commit's library kept under tools/_probe/
LIB_PATH = os.environ.get("COBEVT_HIP_LIB") or LIB_PATH

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_long_p = ctypes.POINTER(ctypes.c_long)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/cobevt_hip.h
SIGNATURES = {
    "cobevt_abi_version": (ctypes.c_int, []),
    "cobevt_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "cobevt_conv2d_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv3x3_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_basicblock_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_dsblock_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv3x3_wfrag_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv3x3_ds_wfrag_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_stem_conv7x7s2_pool": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_stem_conv7x7s2": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_stem_conv7x7s2_pool_u8": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_bev_embed_linear_rows": (ctypes.c_int, [_vp] * 9 + [_c_long_p, ctypes.c_float, _vp]),
    "cobevt_linear_rows_wfrag": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_long_p, ctypes.c_float, _vp]),
    "cobevt_linear_rows": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_long_p, ctypes.c_float, _vp]),
    "cobevt_attn_mlp_chain": (ctypes.c_int, [_vp] * 14 + [_c_int_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp]),
    "cobevt_pairwise_warp": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_float, _vp]),
    "cobevt_pairwise_warp_bwd": (ctypes.c_int, [_vp] * 4 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_float, _vp]),
    "cobevt_agent_message_reduce": (ctypes.c_int, [_vp] * 5 + [ctypes.c_int] * 6 + [_vp]),
    "cobevt_gru_zero_state": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_long, ctypes.c_int, _vp]),
    "cobevt_agent_softmax_sum": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp] + [ctypes.c_int] * 6 + [_vp]),
    "cobevt_window_attention": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int_p, ctypes.c_float, _vp]),
    "cobevt_window_attention_ksplit": (ctypes.c_int, [_vp] * 8 + [_c_int_p, ctypes.c_float, ctypes.c_int, ctypes.c_long, _vp]),
    "cobevt_window_attention_lse": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int_p, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_uint, _vp, _vp]),
    "cobevt_conv3x3_head_nchw": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 6 + [_vp]),
    "cobevt_weighted_cross_entropy_bwd": (ctypes.c_int, [_vp] * 6 + [ctypes.c_int] * 3 + [_vp]),
    "cobevt_attention_dropout_mask": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_uint, _vp, _vp]),
    "cobevt_layernorm_bwd": (ctypes.c_int, [_vp] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp]),
    "cobevt_conv_wgrad": (ctypes.c_int, [_vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv_wgrad_blocked": (ctypes.c_int, [_vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_gelu": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_long, _vp]),
    "cobevt_layernorm_fwd_t": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _c_int_p, _vp]),
    "cobevt_layernorm_bwd_t": (ctypes.c_int, [_vp] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, _c_int_p, _vp]),
    "cobevt_gelu_bf16": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_long, _vp]),
    "cobevt_window_attention_bwd": (ctypes.c_int, [_vp] * 13 + [_c_int_p, ctypes.c_float, ctypes.c_float, ctypes.c_uint, _vp, _vp]),
    "cobevt_layernorm": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int, _vp]),
    "cobevt_fax_ray_embed": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, _vp]),
    "cobevt_fax_bev_embed": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_maxpool3x3s2": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           _vp]),
    "cobevt_to_nhwc": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, _c_long_p, _vp]),
    "cobevt_from_nhwc": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _c_long_p, _vp]),
    "cobevt_regroup": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, _vp]),
    "cobevt_invert_small": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_resize_nhwc": (ctypes.c_int, [_vp, _vp] + [ctypes.c_int] * 8 + [_vp]),
    "cobevt_channel_affine": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_long, ctypes.c_int, ctypes.c_long, _vp]),
    "cobevt_sttf_warp": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _vp]),
    "cobevt_softmax_argmax": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_seg_class_counts": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_weighted_cross_entropy": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_linear_rows_small_k": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_long_p, ctypes.c_float, _vp]),
    "cobevt_bev_embed_linear_rows_small_k": (ctypes.c_int, [_vp] * 9 + [_c_long_p, ctypes.c_float, _vp]),
    "cobevt_sigmoid_focal_loss": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, _vp]),
    "cobevt_iou_counts": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_agent_max": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, _vp]),
    "cobevt_bottleneck_nhwc": (ctypes.c_int, [_vp] * 8 + [_c_int_p, _vp]),
    "cobevt_bottleneck_f32_nhwc": (ctypes.c_int, [_vp] * 9 + [_c_int_p, _vp]),
    "cobevt_attention_index_map": (ctypes.c_int, [_c_int_p, ctypes.c_int, _vp, _vp]),
    "cobevt_attention_bias_index": (ctypes.c_int, [_c_int_p, _c_int_p, ctypes.c_int, _vp, _vp]),
    "cobevt_depthwise_conv_nhwc": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_spatial_mean_nhwc": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_se_gate": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_mean_linear_rows_small_k": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_long_p, ctypes.c_float, _vp]),
    "cobevt_proj_chain": (ctypes.c_int, [_vp] * 10 + [_c_int_p, ctypes.c_float, _vp]),
    "cobevt_proj_chain_kv": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_void_p), _c_int_p, ctypes.c_float, _vp]),
    "cobevt_swap_fusion_stage": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int_p,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp]),
    "cobevt_channel_sums": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, _vp]),
    "cobevt_f64_to_f32": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "cobevt_bn_batch_stats": (ctypes.c_int, [_vp] * 10 + [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "cobevt_bn_finalize": (ctypes.c_int, [_vp] * 10 + [ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "cobevt_bn_apply": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_bn_backward": (ctypes.c_int, [_vp] * 9 + [ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, _vp]),
    "cobevt_maxpool3x3s2_bwd": (ctypes.c_int, [_vp, _vp, _vp] + [ctypes.c_int] * 5 + [_vp]),
    "cobevt_maxpool3x3s2_bwd_t": (ctypes.c_int, [_vp, _vp, _vp] + [ctypes.c_int] * 5 + [_vp]),
    "cobevt_fax_bev_query_train": (ctypes.c_int, [_vp] * 6 + [_c_int_p, _vp]),
    "cobevt_fax_bev_query_train_bwd": (ctypes.c_int, [_vp] * 9 + [_c_int_p, _vp]),
    "cobevt_group_mean": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_long, ctypes.c_int, _vp]),
    "cobevt_pixel_unshuffle2_nhwc": (ctypes.c_int, [_vp, _vp] + [ctypes.c_int] * 6 + [_vp]),
    "cobevt_upsample_nearest2_nhwc": (ctypes.c_int, [_vp, _vp] + [ctypes.c_int] * 6 + [_vp]),
    "cobevt_sttf_warp_bwd": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_float, _vp]),
    "cobevt_conv_weight_rows": (ctypes.c_int, [_vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv3_weight_operands": (ctypes.c_int, [_vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv3_weight_operands2": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_void_p), _c_int_p, _vp]),
    "cobevt_linear_weight_frags": (ctypes.c_int, [_vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_conv_wgrad3_chunks": (ctypes.c_int, [_c_int_p]),
    "cobevt_conv_wgrad3": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_linear_wgrad_chunks": (ctypes.c_int, [_c_long_p]),
    "cobevt_linear_wgrad": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_long_p, _vp]),
    "cobevt_wgrad_block_operand": (ctypes.c_int, [_vp, _vp, _c_int_p, _vp]),
    "cobevt_swish": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_long, _vp]),
    "cobevt_depthwise_wgrad": (ctypes.c_int, [_vp, _vp, _vp, _c_int_p, _vp]),
    "cobevt_resize_bilinear_bwd": (ctypes.c_int, [_vp, _vp] + [ctypes.c_int] * 7 + [_vp]),
    "cobevt_sigmoid_focal_loss_bwd": (ctypes.c_int, [_vp] * 7 + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_float, ctypes.c_int, _vp]),
    "cobevt_peer_window_alloc": (ctypes.c_int, [ctypes.c_long, ctypes.POINTER(_vp), _vp]),
    "cobevt_peer_window_open": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "cobevt_peer_window_close": (ctypes.c_int, [_vp]),
    "cobevt_peer_window_free": (ctypes.c_int, [_vp]),
    "cobevt_peer_window_status": (ctypes.c_int, [_vp, ctypes.c_long, _c_int_p, _c_int_p, _vp]),
    "cobevt_peer_window_status_async": (ctypes.c_int, [_vp, ctypes.c_long, _vp, _vp]),
    "cobevt_calibrate_mfma": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "cobevt_calibrate_copy": (ctypes.c_int, [_vp, _vp, ctypes.c_long, _vp]),
    "cobevt_calibrate_clock_khz": (ctypes.c_int, [_c_int_p, _c_int_p]),
    "cobevt_peer_exchange": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                            _c_int_p, _c_int_p, ctypes.c_long, ctypes.c_long, _vp]),
    "cobevt_host_fetch": (ctypes.c_int, [_vp, _vp, ctypes.c_long, ctypes.c_int, _vp]),
    "cobevt_channel_gate_nhwc": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
}

_libs = {}
# "" = libcobevt_hip.so; "f32s" = libcobevt_hip_f32s.so, the same sources and C ABI built with -DCOBEVT_F32_SPLIT=1: fp32-storage
# kernels on the split-bf16 matrix path (cobevt_amd/build.py, csrc/common.hpp).  host.set_compute_dtype selects it.
# "f32h" = libcobevt_hip_f32h.so (-DCOBEVT_F32_SPLIT=2, round 6): fp32 storage, one fp16 MFMA per 16-byte piece with the weight
# operand as a single fp16 term - half the split-bf16 matrix time at 11-bit weights.  Only the ResNet encoder's convolutions are
# routed through it, and only under host.set_compute_dtype("fp32_fast"): `set_encoder_variant` names the library `encoder_scope()`
# switches to for the launches issued inside it (host/resnet_ms.py).
_variant = ""
_encoder_variant = None
VARIANTS = ("", "f32s", "f32h")
LIB_PATH_F32S = os.environ.get("COBEVT_HIP_LIB_F32S") or os.path.join(_HERE, "csrc", "libcobevt_hip_f32s.so")
LIB_PATH_F32H = os.environ.get("COBEVT_HIP_LIB_F32H") or os.path.join(_HERE, "csrc", "libcobevt_hip_f32h.so")


class CobevtHipError(RuntimeError):
    pass


def set_variant(variant):
    global _variant
    if variant not in VARIANTS:
        raise CobevtHipError("unknown library variant %r" % (variant,))
    _variant = variant


def get_variant():
    return _variant


def set_encoder_variant(variant):
    """The library the ResNet encoder's launches use instead of the active one (None: no override)."""
    global _encoder_variant
    if variant is not None and variant not in VARIANTS:
        raise CobevtHipError("unknown library variant %r" % (variant,))
    _encoder_variant = variant


def get_encoder_variant():
    return _encoder_variant


class encoder_scope(object):
    """with lib.encoder_scope(): ...  - launches issued inside go to the encoder's library variant, when one is set."""

    def __enter__(self):
        global _variant
        self.prev = _variant
        if _encoder_variant is not None:
            _variant = _encoder_variant
        return self

    def __exit__(self, *exc):
        global _variant
        _variant = self.prev
        return False


def load(variant=None):
    """Load (once per variant) and return the ctypes handle of the active library; raises if the HIP extension has not been built."""
    v = _variant if variant is None else variant
    lib = _libs.get(v)
    if lib is not None:
        return lib
    path = {"f32s": LIB_PATH_F32S, "f32h": LIB_PATH_F32H}.get(v, LIB_PATH)
    if not os.path.exists(path):
        raise CobevtHipError(
            "%s not found at %s — build it with `python -m cobevt_amd.build` "
            "(there is no CPU / eager fallback for the hot path)" % (os.path.basename(path), path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    if lib.cobevt_abi_version() != 1:
        raise CobevtHipError("%s ABI version mismatch" % os.path.basename(path))
    _libs[v] = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cobevt_strerror(rc).decode()
        raise CobevtHipError("%s failed: %s (code %d)" % (what, msg, rc))
