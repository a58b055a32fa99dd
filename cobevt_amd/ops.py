"""Thin tensor-level wrappers over the C ABI (include/cobevt_hip.h) + host-side weight preparation.

torch is used for device memory, streams and one-time weight folding only; every arithmetic op of the
forward path is a HIP kernel launched through ctypes.  All functions require CUDA (ROCm) tensors and raise
otherwise — there is no CPU path here (the CPU restatement lives in oracle/ and is test infrastructure).
"""
import ctypes
import os
import math

import torch

from . import lib as _L
from .lib import CobevtHipError

BF16, FP32 = 0, 1
USE_CONV3X3 = True   # route eligible 3x3 convs to the LDS-patch kernel (tests flip this to cover both paths)
USE_CONV3_S2 = True      # 3x3 / stride-2 convs through the strip kernel instead of the generic implicit GEMM
USE_CONV3_SMALL_TILES = True   # grids below the CU count: 32-cout tiles in four-wave workgroups of one / two strips (conv3_tiling)
USE_CONV3_DS = True      # layer3.0 / layer4.0: the projection shortcut inside conv2's launch (cobevt_conv3x3_ds_wfrag_nhwc)
USE_CONV3_WFRAG = True   # ... and, where the fragment-ordered weight table exists, to the barrier-free-per-tap variant
CONV3_VARIANT = 0    # 0 = automatic tile choice of cobevt_conv3x3_wfrag_nhwc; >0 pins one (tools/conv_probe.py)
USE_GEMM_ROWS = True  # route 1x1 stride-1 convs / linears to the dense-row GEMM with fused LayerNorm
USE_GEMM_ROWS2 = False  # dense-row GEMMs through the persistent fragment-ordered kernel (cobevt_linear_rows_wfrag):
USE_GEMM_ROWS3 = True     # K <= 128 bf16 dense rows on the row-chain-style kernel (gemm_rows3.hip)
GEMM_ROWS3_MIN_M = 0
GEMM_ROWS3_MAX_K = 512     # 128: only single-tile rows take the 32-row kernel
GEMM_ROWS3_STRIDED = 1     # stride-2 1x1 convs (ResNet down-sampling) too
GEMM_ROWS3_ROWS64_MIN_M = 0   # > 0: launches with at least this many rows use 64-row workgroups
# measured 3-20 % SLOWER than two independent workgroups per CU (196 VGPRs -> one workgroup per CU, only one 32-KB tile
# of loads in flight per CU: latency-bound on HBM); kept for the parity tests and as the base for a deeper prefetch
USE_STEM_POOL = True  # stem conv + max pool in one launch (the stem map never reaches HBM)
USE_STEM = True       # 7x7/s2 image stem through the space-to-depth kernel instead of the generic small-Cin igemm
ROW_CHAIN_ROWS = 0     # rows per workgroup of the fused row chain: 0 = default (32), 64
USE_ROW_CHAIN = True  # fuse out-proj + skip + pre-norm MLP (+ post-norm) after attention into one launch (bf16)
USE_BOTTLENECK_F32 = True  # the FAX Bottleneck(128, 32) as one launch in fp32 storage (csrc/bottleneck_f32.hip; round 6)
USE_GEMM_ROWS3_F32 = True  # dense-row GEMMs of the fp32 modes on 32-row workgroups with fragment-ordered weights (csrc/gemm_rows3_f32.hip; round 6)
USE_ROW_CHAIN_F32 = True  # ... and its fp32-storage form for C = 128 / hidden 256 (csrc/row_chain_f32.hip; round 6)
BASICBLOCK_TILE_ROWS = 0   # 0 = kernel default; 8 | 16 pins the output tile height (tools/bb_probe.py)
USE_BASICBLOCK = True  # stride-1 BasicBlocks on 64 / 128 channels as one launch (intermediate map stays in LDS)
BASICBLOCK_MAX_C = 128   # widest stride-1 BasicBlock run as one launch (A/B: 64 = the 128-channel blocks of layer 2 as two strip launches)
USE_DSBLOCK = True  # layer2's stride-2 BasicBlock with its projection shortcut as one launch (bf16)
USE_EMBED_GEMM = False  # compute the BEV query embedding inside the to_q GEMM instead of materialising the query:
USE_EMBED_GEMM3 = True    # the BEV query produced inside the launch that projects it.  128 -> 128 with LayerNorm (every OPV2V level): the
                          # wave-level kernel of bev_query.hip (rows in registers, weights in LDS); other widths keep the embedding
                          # kernel + GEMM - inside the 32-row GEMM of gemm_rows3.hip the staging threads become VALU-bound (3 % slower)
# measured SLOWER on MI355X (119 vs 92 us on the level-0 shape, 361 vs 364 frames/s): the producer's 64 LDS
# coefficient reads + ~400 VALU per thread cost more than the 170 MB of HBM traffic they save; kept for parity tests
USE_CHAIN_NEXT = True  # ... and let the row-local GEMM that consumes its output next ride in the same launch
USE_HEAD_CONV = True    # 3x3 convs with <= 4 output channels to NCHW fp32 logits (BevSegHead) on the direct kernel
USE_PROJ_CHAIN = True   # FAX key / value side at 128 feature channels: BN -> ReLU -> 1x1 conv (+ ray embedding) -> LayerNorm -> to_k | to_v
                        # of both attentions in ONE launch per operand, the key / value map itself never reaches HBM (row_chain.hip)
USE_PROJ_CHAIN_KV = True     # FAX key AND value side of a level with 256 / 384 / 512 feature channels in ONE launch (proj_chain_k.hip) instead of four
USE_PROJ_CHAIN_WAVE = True   # ... on maps of >= 32768 rows as independent waves with the rows in registers and the weights in LDS (proj_chain128.hip)
USE_SWAP_STAGE = True   # a SwapFusionBlock half (attention + row chain + next to_qkv) as one launch (swap_stage.hip)
USE_BOTTLENECK = True   # FAX ResNetBottleNeck (128 -> 32 -> 32 -> 128) as one launch (bottleneck.hip) instead of three
ATTN_VARIANT = 0    # 0 = automatic (K/V-resident attention kernel where it applies), 1 = always the streaming kernel, 2 = ... with 64-key tiles (A/B runs)
USE_ATTN_BIG_RESIDENT = False  # plain bf16 windows of 513 .. 1024 keys on the K/V-resident kernel (one launch) instead of key split + merge:
                               # built and measured neutral (659.0 / 659.7 / 668.2 vs 659.2 / 660.3 / 660.5 frames/s, one frame 1.900 vs 1.904 ms,
                               # profiles/r06_attn_big_resident_ab.txt) - kept as an opt-in switch, the key split stays the default
ATTN_KSPLIT = 2     # streaming attention on a small grid with >= 1024 keys (FAX level 2 / global attention): share the keys of a window out
                    # over this many workgroups per query tile + a merge pass (0 / 1 = off).  Round 3 measured it neutral (0 / 2 / 4 =
                    # 582 / 578 / 569 frames/s, profiles/r03_ab_key_split.txt) and left it off; with the round-5 pipeline 2 is a small
                    # consistent gain: 0 / 2 / 4 = 590.9 / 598.3 / 593.0 and 591.4 / 595.2 / 593.4 frames/s, one frame 2.037 / 2.033 /
                    # 2.037 ms (profiles/r05_ab_same_job.txt); parity-tested (tests/test_kernels_gpu.py::test_attention_key_split_matches_single_pass)
ATTN_QSPLIT = 0     # 0 = automatic query split of the resident attention kernel; > 0 pins it (tools/attn_probe.py)


def dcode(dtype):
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float32:
        return FP32
    raise CobevtHipError("compute dtype must be torch.bfloat16 or torch.float32, got %s" % dtype)


def _apply_env_flags():
    """COBEVT_FLAGS="USE_EMBED_GEMM=0,CONV3_VARIANT=150": A/B switches for bench runs (every flag names a HIP path)."""
    for item in os.environ.get("COBEVT_FLAGS", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            k = k.strip()
            if k not in globals() or not (k.startswith("USE_") or k in ("CONV3_VARIANT", "BASICBLOCK_TILE_ROWS", "ROW_CHAIN_ROWS", "GEMM_ROWS3_MIN_M", "GEMM_ROWS3_MAX_K", "GEMM_ROWS3_STRIDED", "GEMM_ROWS3_ROWS64_MIN_M", "BASICBLOCK_MAX_C", "ATTN_VARIANT", "ATTN_QSPLIT", "ATTN_KSPLIT")):
                raise CobevtHipError("COBEVT_FLAGS: unknown switch %r" % k)
            globals()[k] = int(v) if not k.startswith("USE_") else bool(int(v))


_apply_env_flags()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise CobevtHipError("cobevt_amd kernels need tensors on a ROCm device (got a CPU tensor); "
                                 "there is no CPU fallback")


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


# ----------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py roofline leg): HIP events on the launch stream around each C-ABI call
# ----------------------------------------------------------------------------------------------
_PROFILE = None


class LaunchProfile(object):
    """with LaunchProfile() as prof: model(batch) -> prof.summary(): {family: {calls, ms, flops, bytes}}"""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _PROFILE
        _PROFILE = self
        return self

    def __exit__(self, *exc):
        global _PROFILE
        _PROFILE = None

    def summary(self, by_shape=False):
        torch.cuda.synchronize()
        out = {}
        for fam, flops, nbytes, e0, e1 in self.records:
            if not by_shape:
                fam = fam.split("|")[0]
            d = out.setdefault(fam, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


class _Timed(object):
    __slots__ = ("fam", "flops", "nbytes", "e0")

    def __init__(self, fam, flops, nbytes):
        self.fam, self.flops, self.nbytes = fam, flops, nbytes

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        _PROFILE.records.append((self.fam, self.flops, self.nbytes, self.e0, e1))


class _NoTime(object):
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


_NOTIME = _NoTime()


def _timed(fam, flops_fn):
    """flops_fn() -> (algorithmic flops, algorithmic bytes); only evaluated while profiling."""
    if _PROFILE is None:
        return _NOTIME
    f, b = flops_fn()
    return _Timed(fam, f, b)


# ----------------------------------------------------------------------------------------------
# weight preparation (host, once per module/dtype)
# ----------------------------------------------------------------------------------------------
def bn_affine(bn):
    """Eval-mode BatchNorm as per-channel (scale, shift) in float64."""
    g = bn.weight.detach().double() if bn.weight is not None else torch.ones_like(bn.running_var).double()
    b = bn.bias.detach().double() if bn.bias is not None else torch.zeros_like(bn.running_var).double()
    s = g / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return s, b - bn.running_mean.detach().double() * s


class ConvPlan(object):
    """A conv / linear layer lowered to the implicit-GEMM kernel: folded, re-laid-out weights + static params."""

    def __init__(self, weight, bias=None, bn=None, pre_bn=None, pre_relu=False, stride=1, pad=0, act=0,
                 upsample=False, store_mode=0, dtype=torch.bfloat16, device="cuda", smallc=False, ln=None,
                 ln_folded_eps=None, pad_br=None):
        """ln = nn.LayerNorm-like (weight, bias, eps) applied to the input rows of a Linear / 1x1 layer: its affine
        is folded here (W' = W diag(gamma), b' = b + W beta) so the kernels only have to normalise."""
        w = weight.detach().double().cpu()
        if w.dim() == 2:
            w = w[:, :, None, None]
        cout, cin, kh, kw = w.shape
        # pad = zero rows / columns before the first input pixel (top, left); pad_br = after the last one (bottom, right) when
        # it differs - TensorFlow-"same" padding of the EfficientNet stem.  Only the generic implicit GEMM takes the
        # asymmetric form (its gather bounds-checks every tap).
        asym = pad_br is not None and int(pad_br) != int(pad)
        self.pad_br = int(pad if pad_br is None else pad_br)
        b = bias.detach().double().cpu() if bias is not None else torch.zeros(cout, dtype=torch.float64)
        has_bias = bias is not None or bn is not None
        self.has_ln, self.ln_eps = False, 0.0
        if ln_folded_eps is not None:        # the caller already folded a LayerNorm affine into weight / bias: normalise only
            if ln is not None or kh != 1 or kw != 1 or bn is not None or pre_bn is not None:
                raise CobevtHipError("ln_folded_eps applies to plain Linear / 1x1 layers without another ln")
            self.has_ln, self.ln_eps = True, float(ln_folded_eps)
        if ln is not None:
            if kh != 1 or kw != 1 or bn is not None or pre_bn is not None:
                raise CobevtHipError("LayerNorm folding applies to plain Linear / 1x1 layers")
            g, be = ln.weight.detach().double().cpu(), ln.bias.detach().double().cpu()
            b = b + (w[:, :, 0, 0] @ be)
            w = w * g[None, :, None, None]
            has_bias = True
            self.has_ln, self.ln_eps = True, float(ln.eps)
        if bn is not None:
            s, sh = bn_affine(bn)
            s, sh = s.cpu(), sh.cpu()
            w = w * s[:, None, None, None]
            b = b * s + sh
        self.dtype = dtype
        self.code = dcode(dtype)
        bke = 32 if self.code == BF16 else 16
        ch = 8 if self.code == BF16 else 4
        K = kh * kw * cin
        kpad = (K + bke - 1) // bke * bke
        wk = torch.zeros(cout, kpad, dtype=torch.float64)
        wk[:, :K] = w.permute(0, 2, 3, 1).reshape(cout, K)
        self.wgt = wk.to(torch.float32).to(dtype).to(device).contiguous()
        self.bias = b.to(torch.float32).to(device).contiguous() if has_bias else None
        self.pre_scale = self.pre_shift = None
        if pre_bn is not None:
            s, sh = bn_affine(pre_bn)
            self.pre_scale = s.to(torch.float32).to(device).contiguous()
            self.pre_shift = sh.to(torch.float32).to(device).contiguous()
        self.pre_relu = int(bool(pre_relu))
        self.klut = None
        self.smallc = int(bool(smallc))
        if not smallc and cin % ch != 0:
            raise CobevtHipError("Cin=%d must be a multiple of %d for the %s kernel (use smallc=True for image "
                                 "inputs)" % (cin, ch, dtype))
        if smallc:
            k = torch.arange(kpad)
            tap, c = k // cin, k % cin
            code = ((tap // kw) << 20) | ((tap % kw) << 10) | c
            code[k >= K] = -1
            self.klut = code.to(torch.int32).to(device).contiguous()
        # 3x3 / stride 1 / pad 1 fast path (conv3x3.hip): weights [Cout][Cin/cc][9][cc]
        self.wgt3, self.cc3 = None, 0
        if kh == 3 and kw == 3 and int(stride) in (1, 2) and int(pad) == 1 and not asym and int(act) <= 2 and not smallc and pre_bn is None \
                and int(store_mode) in (0, 1) and not (int(stride) == 2 and (upsample or int(store_mode) != 0)):
            cands = (64, 32) if self.code == BF16 else (32, 16)
            for cc in cands:
                if cin % cc == 0:
                    w3 = w.permute(0, 2, 3, 1).reshape(cout, 9, cin // cc, cc).permute(0, 2, 1, 3)
                    if int(stride) == 1:          # the LDS-staged kernel is stride 1 only
                        self.wgt3 = w3.to(torch.float32).to(dtype).to(device).contiguous()
                    self.cc3 = cc
                    break
        # the same weights in MFMA B-fragment order for cobevt_conv3x3_wfrag_nhwc (128-byte chunks only):
        # [Cout_p/32][Cin/cc][9][KG][lane = 32*half + cout%32][16 bytes], Cout zero-padded to a multiple of 128
        self.wfrag, self.coutp3 = None, 0
        if self.cc3 and self.cc3 == (64 if self.code == BF16 else 32):
            cc, coutp = self.cc3, (cout + 127) // 128 * 128
            wp = torch.zeros(coutp, cin // cc, 9, cc, dtype=torch.float64)
            wp[:cout] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin // cc, cc).permute(0, 2, 1, 3)
            wf = wp.reshape(coutp // 32, 32, cin // cc, 9, 4, 2, ch).permute(0, 2, 3, 4, 5, 1, 6)
            self.wfrag = wf.to(torch.float32).to(dtype).to(device).contiguous()
            self.coutp3 = coutp
        # ResNet stem fast path (stem7x7.hip): 7x7 / stride 2 / pad 3 on 3 channels as a 4x4 conv on the 2x2
        # space-to-depth image; W'[n][a][b][dy][dx][c] = w[n][c][2a+dy-1][2b+dx-1]
        self.wgt_stem = None
        if smallc and kh == 7 and kw == 7 and int(stride) == 2 and int(pad) == 3 and not asym and int(act) <= 2 and cin == 3 and pre_bn is None \
                and int(store_mode) == 0 and not upsample and cout % (8 if self.code == BF16 else 4) == 0:
            ws = torch.zeros(cout, 4, 4, 16, dtype=torch.float64)
            for a in range(4):
                for bb in range(4):
                    for dy in range(2):
                        for dx in range(2):
                            ih, iw = 2 * a + dy - 1, 2 * bb + dx - 1
                            if 0 <= ih < 7 and 0 <= iw < 7:
                                ws[:, a, bb, dy * 6 + dx * 3:dy * 6 + dx * 3 + 3] = w[:, :, ih, iw]
            self.wgt_stem = ws.reshape(cout, 256).to(torch.float32).to(dtype).to(device).contiguous()
        # dense-row GEMM fast path (gemm_rows.hip) for 1x1 / stride 1: weights [Cout][K rounded to a 256-byte tile]
        self.wgt_rows, self.kp_rows, self.wfrag_rows = None, 0, None
        if kh == 1 and kw == 1 and int(stride) in (1, 2) and int(pad) == 0 and not asym and not smallc and int(store_mode) == 0 \
                and not upsample:
            tk = 128 if self.code == BF16 else 64
            kp = (K + tk - 1) // tk * tk
            wr = torch.zeros(cout, kp, dtype=torch.float64)
            wr[:, :K] = w.reshape(cout, K)
            self.wgt_rows = wr.to(torch.float32).to(dtype).to(device).contiguous()
            self.kp_rows = kp
            # the same matrix in MFMA fragment order for the fused row chain (row_chain.hip):
            # [N_p/32 tiles][kp/16 k-groups][lane = 32*half + n%32][8 bf16], N zero-padded to a multiple of 128
            eg = 8 if self.code == BF16 else 4          # elements per 16 bytes
            npad = (cout + 127) // 128 * 128
            wp_ = torch.zeros(npad, kp, dtype=torch.float64)
            wp_[:cout] = wr
            wf = wp_.reshape(npad // 32, 32, kp // (2 * eg), 2, eg).permute(0, 2, 3, 1, 4)
            self.wfrag_rows = wf.to(torch.float32).to(dtype).to(device).contiguous()
        # BevSegHead-style 3x3 convs with 1..4 output channels to fp32 NCHW logits: a direct kernel (postprocess.hip)
        self.wgt_head = None
        if kh == 3 and kw == 3 and int(stride) == 1 and int(pad) == 1 and not asym and not smallc and int(store_mode) == 2 and int(act) == 0 \
                and not upsample and pre_bn is None and cout <= 4 and cin * (2 if self.code == BF16 else 4) <= 256 and cin % ch == 0:
            self.wgt_head = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(torch.float32).to(device).contiguous()
        self.cin, self.cout, self.kh, self.kw = cin, cout, kh, kw
        self.K, self.kpad = K, kpad
        self.stride, self.pad, self.act = int(stride), int(pad), int(act)
        self.upsample = int(bool(upsample))
        self.store_mode = int(store_mode)

    def out_hw(self, h, w):
        hv, wv = (2 * h, 2 * w) if self.upsample else (h, w)
        return ((hv + self.pad + self.pad_br - self.kh) // self.stride + 1,
                (wv + self.pad + self.pad_br - self.kw) // self.stride + 1)


def ln_fusable(plan):
    """The row normalisation can run inside the dense-row GEMM when the row fits one 256-byte K-tile."""
    return USE_GEMM_ROWS and plan.wgt_rows is not None and plan.K <= (128 if plan.code == BF16 else 64)


def conv3_tiling(n, ho, wo, cin, cout, cc, cus=256, stride=1, bf16=True, packed=0):
    """Pick the 3x3 kernel / tile shape for one launch: 0 = the LDS-staged kernel (cobevt_conv3x3_nhwc), else the
    `variant` of cobevt_conv3x3_wfrag_nhwc (100 + 10*MT + bn64: MT strips of 2x16 pixels x 128|64 couts per workgroup).
    The LDS-staged kernel runs two workgroups per CU and wins once its grid is >= 2 per CU; below that the kernel time
    is whole workgroup lifetimes, so the strip count MT is chosen to make the grid a whole number of waves of `cus`
    workgroups (cycle model: 12k fixed + 40 cycles per MFMA-tile-step, both from the s_memtime traces)."""
    if stride == 1:
        if cout <= 32 and bf16 and cc == 64:
            # 32-cout tiles in four-wave workgroups, two per CU (a 64-cout tile would waste half its MFMAs): one round of
            # 2 * cus workgroups if the strips allow it, else the fewest strips per workgroup (tools/conv_graph_probe.py)
            nstrips = n * (-(-ho // 2)) * (-(-wo // 16))
            best = None
            for mt in ((1, 2, 3, 4, 5) if USE_CONV3_SMALL_TILES else (3, 4, 5)):
                blocks = -(-nstrips // mt) * -(-cout // 32)
                cost = -(-blocks // (2 * cus)) * (6000 + 9 * (cin // cc) * mt * 80)
                if best is None or cost < best[0]:
                    best = (cost, 100 + 10 * mt + 3)
            return best[1]
        if cout < 64:
            return 0
        th, bn = (16, 64) if cout <= 64 else (8, 128)
        blocks_old = n * (-(-ho // th)) * (-(-wo // 16)) * (-(-cout // bn))
        if blocks_old >= 2 * cus and not (packed and cout >= 128):      # (the LDS-staged kernel has no packed form)
            return 0
    nstrips = n * (-(-ho // 2)) * (-(-wo // 16))
    nsteps = 9 * (cin // cc)
    best = None
    for bn64 in ((1,) if cout <= 64 else (0, 1)):
        tile_n = 64 if bn64 else 128
        for mt in (3, 4, 5, 6):
            blocks = -(-nstrips // mt) * -(-cout // tile_n)
            # packed (third library, fp32 storage): the 128-cout tiles' waves own two k-groups per tap and issue ONE MFMA for both
            # (csrc/common.hpp kXPack); the 64-cout tiles' waves own one and cannot
            # (second library: three MFMAs per k-group pair instead of four on those tiles, kXPack3 - packed = 0.75)
            per_step = int(40 * (float(packed) if packed else 1.0)) if (packed and not bn64) else 40
            cost = -(-blocks // cus) * (12000 + nsteps * mt * (tile_n // 32) * per_step
                                       + (1500 * mt * (cin // cc) if stride == 2 else 0))   # exposed patch refills
            if best is None or cost < best[0]:
                best = (cost, 100 + 10 * mt + bn64, blocks)
    if USE_CONV3_SMALL_TILES and stride == 1 and bf16 and cc == 64 and not packed and best[2] < cus:
        # a grid that does not reach the CU count (one image: the decoder maps, the small FAX maps): the launch lasts one workgroup
        # lifetime, so the shortest workgroup wins - 32-cout tiles in four-wave workgroups with one or two strips each (round 6)
        for mt in (1, 2, 3, 4, 5):
            blocks = -(-nstrips // mt) * -(-cout // 32)
            cost = -(-blocks // (2 * cus)) * (6000 + nsteps * mt * 80)
            if cost < best[0]:
                best = (cost, 100 + 10 * mt + 3, blocks)
    return best[1]


def conv2d(x, plan, residual=None, out=None):
    """x: (N,H,W,Cin) channels-last contiguous in plan.dtype (fp32 image for smallc plans).  Returns the output
    in the layout selected by plan.store_mode.  `out` may be a pre-zeroed, spatially larger (N,Hp,Wp,Cout) map.
    Plans built with ln=... normalise the rows of x first (LayerNorm affine already folded into the weights)."""
    _need_cuda(x, residual, out)
    ln = plan.has_ln
    if ln and not ln_fusable(plan):
        x = layernorm(x, None, None, plan.ln_eps)
        ln = False
    n, h, w, cin = x.shape
    if cin != plan.cin or not x.is_contiguous():
        raise CobevtHipError("conv2d: bad input %s (contiguous=%s) for Cin=%d" % (tuple(x.shape), x.is_contiguous(), plan.cin))
    if x.dtype != (torch.float32 if plan.smallc else plan.dtype):
        raise CobevtHipError("conv2d: input dtype %s does not match plan (%s, smallc=%d)" % (x.dtype, plan.dtype, plan.smallc))
    ho, wo = plan.out_hw(h, w)
    sm = plan.store_mode
    out_h, out_w = ho, wo
    if out is None:
        if sm == 0:
            out = torch.empty((n, ho, wo, plan.cout), device=x.device, dtype=plan.dtype)
        elif sm == 1:
            out = torch.empty((n, ho // 2, wo // 2, plan.cout * 4), device=x.device, dtype=plan.dtype)
        elif sm == 2:
            out = torch.empty((n, plan.cout, ho, wo), device=x.device, dtype=torch.float32)
        else:
            out = torch.empty((n, ho, wo, plan.cout), device=x.device, dtype=torch.float32)
    else:
        if sm not in (0, 3) or out.dim() != 4 or out.shape[0] != n or out.shape[3] != plan.cout or not out.is_contiguous():
            raise CobevtHipError("conv2d: incompatible `out`")
        out_h, out_w = out.shape[1], out.shape[2]
        if out_h < ho or out_w < wo:
            raise CobevtHipError("conv2d: `out` smaller than the convolution result")
    if residual is not None:
        if tuple(residual.shape) != (n, ho, wo, plan.cout) or residual.dtype != plan.dtype or not residual.is_contiguous():
            raise CobevtHipError("conv2d: residual must be (N,Ho,Wo,Cout) contiguous in the compute dtype")
    def cost():
        m = n * ho * wo
        esz = 2 if plan.code == BF16 else 4
        nbytes = x.numel() * x.element_size() + plan.cout * plan.K * esz + out.numel() * out.element_size()
        if residual is not None:
            nbytes += residual.numel() * esz
        return 2.0 * m * plan.cout * plan.K, float(nbytes)

    if plan.wgt_head is not None and USE_HEAD_CONV and residual is None and (out_h, out_w) == (ho, wo):
        with _timed("head3x3|%d->%d %dx%dx%d" % (cin, plan.cout, n, ho, wo), cost):
            rc = _L.load().cobevt_conv3x3_head_nchw(_p(x), _p(plan.wgt_head), _p(plan.bias), _p(out), plan.code, n, h, w, cin, plan.cout,
                                                    _stream())
        _L.check(rc, "cobevt_conv3x3_head_nchw")
        return out
    if plan.wgt_stem is not None and USE_STEM and h % 2 == 0 and w % 2 == 0 and residual is None and (out_h, out_w) == (ho, wo):
        sdims = _ints([plan.code, n, h, w, plan.cout, plan.act])
        with _timed("stem7x7|%dx%dx%d" % (n, h, w), cost):
            rc = _L.load().cobevt_stem_conv7x7s2(_p(x), _p(plan.wgt_stem), _p(plan.bias), _p(out), sdims, _stream())
        _L.check(rc, "cobevt_stem_conv7x7s2")
        return out
    if plan.wgt_rows is not None and USE_GEMM_ROWS and not (residual is not None and (out_h, out_w) != (ho, wo)):
        ldims = (ctypes.c_long * 16)(plan.code, n * ho * wo, plan.cout, plan.K, plan.kp_rows, cin, plan.pre_relu, plan.act,
                                     ho, wo, out_h, out_w, int(ln), plan.stride, h, w)
        v2 = USE_GEMM_ROWS2 and plan.wfrag_rows is not None and plan.cout % (8 if plan.code == BF16 else 4) == 0
        # K <= 128 bf16 rows: the row-chain-style kernel (32-row workgroups, fragment-ordered weights from L2)
        v3 = (USE_GEMM_ROWS3 and plan.code == BF16 and plan.wfrag_rows is not None and plan.kp_rows <= GEMM_ROWS3_MAX_K
              and plan.K % 8 == 0 and (out_h, out_w) == (ho, wo) and plan.cout % 8 == 0 and plan.cout <= 4096 and sm == 0
              and not (plan.stride > 1 and (residual is not None or not GEMM_ROWS3_STRIDED))
              and not (ln and plan.kp_rows > 128) and n * ho * wo >= GEMM_ROWS3_MIN_M)
        # ... and its fp32-storage form (round 6, csrc/gemm_rows3_f32.hip): K <= 512, the fused LayerNorm for K = 128
        v3 = v3 or (USE_GEMM_ROWS3_F32 and plan.code == FP32 and plan.wfrag_rows is not None and plan.kp_rows <= 512 and plan.K % 4 == 0
                    and (out_h, out_w) == (ho, wo) and plan.cout % 4 == 0 and plan.cout <= 4096 and sm == 0
                    and not (plan.stride > 1 and residual is not None) and not (ln and plan.K != 128)
                    and not (ln and plan.pre_scale is not None) and n * ho * wo >= GEMM_ROWS3_MIN_M)
        if v3:
            d3 = (ctypes.c_long * 14)(plan.code, n * ho * wo, plan.cout, plan.K, cin, plan.pre_relu, plan.act, int(ln),
                                      plan.stride, ho, wo, h, w,
                                      64 if (GEMM_ROWS3_ROWS64_MIN_M and n * ho * wo >= GEMM_ROWS3_ROWS64_MIN_M) else 32)
            with _timed("gemm_rows|%d->%d M=%d%s%s r32" % (cin, plan.cout, n * ho * wo, " ln" if ln else "",
                                                        " s%d" % plan.stride if plan.stride > 1 else ""), cost):
                rc = _L.load().cobevt_linear_rows_small_k(_p(x), _p(plan.wfrag_rows), _p(plan.bias), _p(residual), _p(plan.pre_scale),
                                                          _p(plan.pre_shift), _p(out), d3, ctypes.c_float(plan.ln_eps), _stream())
            _L.check(rc, "cobevt_linear_rows_small_k")
            return out
        with _timed("gemm_rows|%d->%d M=%d%s%s" % (cin, plan.cout, n * ho * wo, " ln" if ln else "",
                                                   " s%d" % plan.stride if plan.stride > 1 else ""), cost):
            if v2:      # persistent workgroups, fragment-ordered weights
                rc = _L.load().cobevt_linear_rows_wfrag(_p(x), _p(plan.wfrag_rows), _p(plan.bias), _p(residual),
                                                        _p(plan.pre_scale), _p(plan.pre_shift), _p(out), ldims,
                                                        ctypes.c_float(plan.ln_eps), _stream())
            else:
                rc = _L.load().cobevt_linear_rows(_p(x), _p(plan.wgt_rows), _p(plan.bias), _p(residual), None, None,
                                                  _p(plan.pre_scale), _p(plan.pre_shift), _p(out), ldims,
                                                  ctypes.c_float(plan.ln_eps), _stream())
        _L.check(rc, "cobevt_linear_rows_wfrag" if v2 else "cobevt_linear_rows")
        return out
    variant = 0
    # (the stride-2 strip variants sit at the 256-VGPR limit in fp32: with the operand split of the f32s library they spill
    # hundreds of registers, so that library's three stride-2 3x3 convs of a ResNet take the generic implicit GEMM)
    if plan.wfrag is not None and (out_h, out_w) == (ho, wo) and USE_CONV3X3 and USE_CONV3_WFRAG \
            and (plan.stride == 1 or (USE_CONV3_S2 and not (plan.code == FP32 and _L.get_variant() == "f32s"))):
        variant = CONV3_VARIANT or conv3_tiling(n, ho, wo, cin, plan.cout, plan.cc3, stride=plan.stride, bf16=plan.code == BF16,
                                                packed=(0.5 if _L.get_variant() == "f32h" else 0.75 if _L.get_variant() == "f32s" else 0) if plan.code == FP32 else 0)
        if variant == 0 and plan.stride == 2:
            variant = 151 if plan.cout <= 64 else 150
    if variant > 0:
        dims = _ints([plan.code, n, h, w, cin, plan.cout, plan.upsample, plan.act, sm, plan.cc3, plan.coutp3, variant,
                      plan.stride])
        with _timed("conv3x3|%d->%d %dx%dx%d" % (cin, plan.cout, n, ho, wo), cost):
            rc = _L.load().cobevt_conv3x3_wfrag_nhwc(_p(x), _p(plan.wfrag), _p(plan.bias), _p(residual), _p(out), dims, _stream())
        _L.check(rc, "cobevt_conv3x3_wfrag_nhwc")
        return out
    if plan.wgt3 is not None and (out_h, out_w) == (ho, wo) and USE_CONV3X3:
        dims = _ints([plan.code, n, h, w, cin, plan.cout, plan.upsample, plan.act, sm, plan.cc3])
        with _timed("conv3x3|%d->%d %dx%dx%d" % (cin, plan.cout, n, ho, wo), cost):
            rc = _L.load().cobevt_conv3x3_nhwc(_p(x), _p(plan.wgt3), _p(plan.bias), _p(residual), _p(out), dims, _stream())
        _L.check(rc, "cobevt_conv3x3_nhwc")
        return out
    dims = _ints([plan.code, n, h, w, cin, ho, wo, plan.cout, plan.kh, plan.kw, plan.stride, plan.pad, plan.K,
                  plan.kpad, plan.upsample, plan.pre_relu, plan.act, sm, out_h, out_w, plan.smallc])
    with _timed("igemm|k%ds%d %d->%d M=%d%s" % (plan.kh, plan.stride, cin, plan.cout, n * ho * wo, " stem" if plan.smallc else ""), cost):
        rc = _L.load().cobevt_conv2d_nhwc(_p(x), _p(plan.wgt), _p(plan.bias), _p(residual), _p(plan.pre_scale),
                                          _p(plan.pre_shift), _p(plan.klut), _p(out), dims, _stream())
    _L.check(rc, "cobevt_conv2d_nhwc")
    return out


def basicblock_fusable(x, plan1, plan2):
    """stride-1 BasicBlock without downsample on 64 / 128 channels: both 3x3 convs in one launch (basicblock.hip)"""
    return (USE_BASICBLOCK and plan1.wfrag is not None and plan2.wfrag is not None and plan1.cin == plan1.cout == plan2.cin
            == plan2.cout and plan1.cout in (64, 128) and plan1.cout <= BASICBLOCK_MAX_C and plan1.act == 1 and plan2.act == 1 and not plan1.upsample
            and not plan2.upsample and plan1.store_mode == 0 and plan2.store_mode == 0 and x.is_contiguous()
            and x.dtype == plan1.dtype and x.numel() < 2 ** 31)


def basicblock(x, plan1, plan2):
    """relu(conv2(relu(conv1(x))) + x) for BN-folded 3x3 plans; x (N,H,W,C) channels-last."""
    _need_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty_like(x)
    dims = _ints([plan1.code, n, h, w, c, BASICBLOCK_TILE_ROWS])

    def cost():
        esz = 2 if plan1.code == BF16 else 4
        return 2.0 * 2 * n * h * w * c * 9 * c, float(2 * x.numel() * esz + 2 * c * 9 * c * esz)

    with _timed("basicblock|%d %dx%dx%d" % (c, n, h, w), cost):
        rc = _L.load().cobevt_basicblock_nhwc(_p(x), _p(plan1.wfrag), _p(plan1.bias), _p(plan2.wfrag), _p(plan2.bias),
                                              _p(out), dims, _stream())
    _L.check(rc, "cobevt_basicblock_nhwc")
    return out


def dsblock_fusable(x, plan1, plan2, plan_ds):
    """layer2's first BasicBlock (64 -> 128, stride 2, 1x1 / stride-2 projection shortcut), bf16: one launch (basicblock.hip)"""
    return (USE_DSBLOCK and x.dtype == torch.bfloat16 and plan1.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous()
            and plan1.wfrag is not None and plan2.wfrag is not None and plan_ds.wfrag_rows is not None
            and (plan1.cin, plan1.cout, plan1.stride) == (64, 128, 2) and (plan2.cin, plan2.cout, plan2.stride) == (128, 128, 1)
            and (plan_ds.cin, plan_ds.cout, plan_ds.stride, plan_ds.kp_rows) == (64, 128, 2, 128)
            and plan1.act == 1 and plan2.act == 1 and plan_ds.act == 0 and plan_ds.pre_scale is None and not plan_ds.has_ln
            and not plan1.upsample and not plan2.upsample and plan1.store_mode == 0 and plan2.store_mode == 0
            and x.shape[3] == 64 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and x.numel() < 2 ** 31)


def dsblock(x, plan1, plan2, plan_ds):
    """relu(conv2(relu(conv1/s2(x))) + ds(x)) for BN-folded plans; x (N,H,W,64) channels-last bf16 -> (N,H/2,W/2,128)."""
    _need_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty((n, h // 2, w // 2, 128), dtype=x.dtype, device=x.device)
    dims = _ints([plan1.code, n, h, w, c, 128])

    def cost():
        px = n * (h // 2) * (w // 2)
        return 2.0 * px * 128 * (9 * 64 + 9 * 128 + 64), float(2 * (x.numel() + out.numel()) + 2 * 128 * (9 * 64 + 9 * 128 + 64))

    with _timed("basicblock|%d->128/s2 %dx%dx%d" % (c, n, h // 2, w // 2), cost):
        rc = _L.load().cobevt_dsblock_nhwc(_p(x), _p(plan1.wfrag), _p(plan1.bias), _p(plan2.wfrag), _p(plan2.bias),
                                           _p(plan_ds.wfrag_rows), _p(plan_ds.bias), _p(out), dims, _stream())
    _L.check(rc, "cobevt_dsblock_nhwc")
    return out


def conv3_ds_fusable(x, plan1, plan2, plan_ds):
    """The down-sampling BasicBlocks of layer3 / layer4 (stride-2 conv1, 1x1 / stride-2 projection shortcut), bf16: the shortcut rides
    in conv2's launch as extra one-tap channel chunks (cobevt_conv3x3_ds_wfrag_nhwc).  Returns the tile variant, or 0."""
    if not (USE_CONV3_DS and USE_CONV3X3 and USE_CONV3_WFRAG and x.dtype == torch.bfloat16 and plan2.dtype == torch.bfloat16
            and x.dim() == 4 and x.is_contiguous() and plan2.wfrag is not None and plan_ds.wgt_rows is not None
            and plan1.stride == 2 and plan2.stride == 1 and plan_ds.stride == 2 and plan2.cc3 == 64
            and plan1.cout == plan2.cin == plan2.cout == plan_ds.cout and plan_ds.cin == x.shape[3] and x.shape[3] % 64 == 0
            and x.shape[3] <= 256 and plan2.cin % 64 == 0 and plan2.act in (0, 1) and plan_ds.act == 0 and plan_ds.pre_scale is None and not plan_ds.has_ln
            and not plan2.upsample and plan2.store_mode == 0 and plan2.bias is not None and plan_ds.bias is not None
            and x.numel() < 2 ** 31):
        return 0
    n, h, w, _ = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    variant = CONV3_VARIANT or conv3_tiling(n, ho, wo, plan2.cin, plan2.cout, 64)
    mt, code = (variant - 100) // 10, (variant - 100) % 10
    return variant if (variant >= 100 and 3 <= mt <= 5 and code in (0, 1)) else 0


def _conv3_ds_table(plan2, plan_ds):
    """conv2's fragment table with the shortcut's one-tap steps appended per 32-cout tile (cached on plan2, keyed by the shortcut plan)"""
    cached = getattr(plan2, "_ds_fused", None)
    if cached is not None and cached[0] is plan_ds:
        return cached[1]
    coutp, cin, cin2 = plan2.coutp3, plan2.cin, plan_ds.cin
    t = coutp // 32
    main = plan2.wfrag.reshape(t, (cin // 64) * 9, 2048)
    wd = torch.zeros((coutp, cin2), dtype=plan2.wfrag.dtype, device=plan2.wfrag.device)
    wd[:plan_ds.cout] = plan_ds.wgt_rows[:, :cin2]
    extra = wd.reshape(t, 32, cin2 // 64, 4, 2, 8).permute(0, 2, 3, 4, 1, 5).reshape(t, cin2 // 64, 2048)
    table = torch.cat([main, extra], dim=1).contiguous()
    plan2._ds_fused = (plan_ds, table)
    return table


def conv3_ds(y, x, plan2, plan_ds, variant):
    """act(conv2(y) + round(ds(x))) for BN-folded plans (the shortcut rounded to the storage type as its own launch would store it): y (N, H/2, W/2, C) = the block's first convolution's output, x (N, H, W, Cin2) its input."""
    _need_cuda(y, x)
    n, ho, wo, c = y.shape
    _, h, w, cin2 = x.shape
    table = _conv3_ds_table(plan2, plan_ds)
    out = torch.empty((n, ho, wo, plan2.cout), dtype=y.dtype, device=y.device)
    dims = _ints([plan2.code, n, ho, wo, c, plan2.cout, plan2.act, plan2.coutp3, variant, h, w, cin2])

    def cost():
        px = n * ho * wo
        return 2.0 * px * plan2.cout * (9 * c + cin2), float(2 * (y.numel() + px * cin2 + out.numel()) + 2 * plan2.cout * (9 * c + cin2))

    with _timed("conv3x3|%d->%d +ds%d %dx%dx%d" % (c, plan2.cout, cin2, n, ho, wo), cost):
        rc = _L.load().cobevt_conv3x3_ds_wfrag_nhwc(_p(y), _p(x), _p(table), _p(plan2.bias), _p(plan_ds.bias), _p(out), dims, _stream())
    _L.check(rc, "cobevt_conv3x3_ds_wfrag_nhwc")
    return out


class BottleneckPlan(object):
    """torchvision Bottleneck(128, 32) lowered for cobevt_bottleneck_nhwc: the three convolutions with their eval BatchNorms
    folded, as MFMA A-operand fragments ([.., 64 lanes, 8 values]: lane = 32 * half + output row, the 8 values are the
    contraction slots 8 * half .. + 7 of a 16-wide k-block).  W3's contraction index is stored in the order the kernel's
    accumulator registers hand conv2's result over: slot 8 * half + j of k-block u <-> mid channel 16u + (j & 3) + 8 (j >> 2) +
    4 * half (bottleneck.hip)."""

    def __init__(self, conv1, bn1, conv2, bn2, conv3, bn3, device="cuda"):
        def fold(conv, bn):
            w = conv.weight.detach().double().cpu()
            s, sh = bn_affine(bn)
            b = conv.bias.detach().double().cpu() * s.cpu() + sh.cpu() if conv.bias is not None else sh.cpu()
            return w * s.cpu()[:, None, None, None], b
        w1, b1 = fold(conv1, bn1)
        w2, b2 = fold(conv2, bn2)
        w3, b3 = fold(conv3, bn3)
        mid, cin = w1.shape[0], w1.shape[1]
        ok = (tuple(w1.shape) == (32, 128, 1, 1) and tuple(w2.shape) == (32, 32, 3, 3) and tuple(w3.shape) == (128, 32, 1, 1)
              and conv1.stride == (1, 1) and conv2.stride == (1, 1) and conv3.stride == (1, 1) and conv2.padding == (1, 1))
        if not ok:
            raise CobevtHipError("the fused Bottleneck kernel is built for Bottleneck(128, 32), stride 1 (got %d -> %d)" % (cin, mid))
        lane = torch.arange(64)
        row, half = lane % 32, lane // 32
        j = torch.arange(8)
        # W1 [8 k-groups][64][8]: W1[row][16 g + 8 half + j]
        g = torch.arange(8)
        k1 = 16 * g[:, None, None] + 8 * half[None, :, None] + j[None, None, :]
        f1 = w1[:, :, 0, 0][row[None, :, None].expand(8, 64, 8), k1]
        # W2 [9 taps][2][64][8]: W2[row][16 u + 8 half + j][ky][kx]
        u = torch.arange(2)
        k2 = 16 * u[:, None, None] + 8 * half[None, :, None] + j[None, None, :]
        w2t = w2.reshape(32, 32, 9)
        f2 = torch.stack([w2t[:, :, t][row[None, :, None].expand(2, 64, 8), k2] for t in range(9)])
        # W3 [4 cout tiles][2][64][8]: W3[32 ct + row][16 u + (j & 3) + 8 (j >> 2) + 4 half]
        k3 = 16 * u[:, None, None] + (j & 3)[None, None, :] + 8 * (j >> 2)[None, None, :] + 4 * half[None, :, None]
        f3 = torch.stack([w3[32 * ct:32 * ct + 32, :, 0, 0][row[None, :, None].expand(2, 64, 8), k3] for ct in range(4)])
        bf = lambda t: t.to(torch.float32).to(torch.bfloat16).to(device).contiguous()      # noqa: E731
        self.w1, self.w2, self.w3 = bf(f1), bf(f2), bf(f3)
        self.b1, self.b2, self.b3 = [t.to(torch.float32).to(device).contiguous() for t in (b1, b2, b3)]


def bottleneck_fusable(x):
    return USE_BOTTLENECK and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] == 128 and x.is_contiguous()


def bottleneck(x, plan, tile_rows=0):
    """relu(conv3(relu(conv2(relu(conv1(x))))) + x) for a BottleneckPlan; x (N, H, W, 128) channels-last bf16."""
    _need_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty_like(x)
    dims = _ints([BF16, n, h, w, c, 32, tile_rows])

    def cost():
        m = n * h * w
        return 2.0 * m * (128 * 32 + 32 * 32 * 9 + 32 * 128), float(2 * x.numel() * 2 + (128 * 32 * 2 + 32 * 32 * 9) * 2)

    with _timed("bottleneck|%dx%dx%d" % (n, h, w), cost):
        rc = _L.load().cobevt_bottleneck_nhwc(_p(x), _p(plan.w1), _p(plan.w2), _p(plan.w3), _p(plan.b1), _p(plan.b2), _p(plan.b3),
                                              _p(out), dims, _stream())
    _L.check(rc, "cobevt_bottleneck_nhwc")
    return out


def bottleneck_f32_fusable(x, p1, p2, p3, y1=None):
    """the fp32-storage Bottleneck(128, 32) as one launch (csrc/bottleneck_f32.hip; round 6): the three separate launches' own plans"""
    return (USE_BOTTLENECK_F32 and x.dtype == torch.float32 and x.dim() == 4 and x.shape[3] == 128 and x.is_contiguous() and x.numel() < 2 ** 31
            and p1.code == FP32 and p1.wfrag_rows is not None and (p1.K, p1.cout, p1.kp_rows, p1.act, p1.stride) == (128, 32, 128, 1, 1)
            and not p1.has_ln and p1.pre_scale is None and p1.bias is not None
            and p2.wfrag is not None and (p2.cin, p2.cout, p2.cc3, p2.stride, p2.act) == (32, 32, 32, 1, 1) and not p2.upsample
            and p2.store_mode == 0 and p2.bias is not None
            and p3.wfrag_rows is not None and (p3.K, p3.cout, p3.act, p3.stride) == (32, 128, 1, 1) and not p3.has_ln and p3.pre_scale is None
            and p3.bias is not None
            and (y1 is None or (y1.dtype == torch.float32 and y1.is_contiguous() and tuple(y1.shape) == tuple(x.shape[:3]) + (32,))))


def bottleneck_f32(x, p1, p2, p3, y1=None):
    """relu(conv3(relu(conv2(relu(conv1(x))))) + x) on (N, H, W, 128) fp32 channels-last; y1 = relu(conv1(x)) when the producer made it."""
    _need_cuda(x, y1)
    n, h, w, c = x.shape
    out = torch.empty_like(x)
    dims = _ints([n, h, w, p3.kp_rows // 8 * 64])

    def cost():
        m = n * h * w
        return 2.0 * m * ((0 if y1 is not None else 128 * 32) + 32 * 32 * 9 + 32 * 128), float((2 * x.numel() + (y1.numel() if y1 is not None else 0)) * 4)

    with _timed("bottleneck|%dx%dx%d%s" % (n, h, w, " y1" if y1 is not None else ""), cost):
        rc = _L.load().cobevt_bottleneck_f32_nhwc(_p(x), _p(y1), _p(p1.wfrag_rows), _p(p1.bias), _p(p2.wfrag), _p(p2.bias),
                                                  _p(p3.wfrag_rows), _p(p3.bias), _p(out), dims, _stream())
    _L.check(rc, "cobevt_bottleneck_f32_nhwc")
    return out


def linear(x, plan, residual=None, out=None):
    """x: (..., K) contiguous tokens -> (..., Cout).  A Linear is the 1x1 case of the implicit GEMM; plans built
    with ln=... apply LayerNorm(x) first (fused into the GEMM when the row fits one K-tile).  out: optional
    preallocated contiguous (..., Cout) result buffer in the compute dtype."""
    lead = x.shape[:-1]
    rows = int(math.prod(lead)) if len(lead) else 1
    x4 = x.reshape(1, 1, rows, x.shape[-1])
    r4 = residual.reshape(1, 1, rows, plan.cout) if residual is not None else None
    o4 = None
    if out is not None:
        if tuple(out.shape) != tuple(lead) + (plan.cout,) or not out.is_contiguous() or out.dtype != x.dtype:
            raise CobevtHipError("linear: `out` must be a contiguous %s tensor of dtype %s" % (tuple(lead) + (plan.cout,), x.dtype))
        o4 = out.reshape(1, 1, rows, plan.cout)
    y = conv2d(x4, plan, residual=r4, out=o4)
    return out if out is not None else y.reshape(*lead, plan.cout)


# ----------------------------------------------------------------------------------------------
def layernorm(x, gamma, beta, eps=1e-5, out=None):
    """LayerNorm over the last axis of a contiguous tensor."""
    _need_cuda(x, gamma, beta)
    if not x.is_contiguous():
        raise CobevtHipError("layernorm: input must be contiguous")
    c = x.shape[-1]
    rows = x.numel() // c
    if out is None:
        out = torch.empty_like(x)
    rc = _L.load().cobevt_layernorm(_p(x), _p(gamma), _p(beta), _p(out), dcode(x.dtype), rows, c, float(eps), 1, 0, 0, 0,
                                    _stream())
    _L.check(rc, "cobevt_layernorm")
    return out


def mean_layernorm(x, gamma, beta, eps=1e-5):
    """x: (B, L, R, C) contiguous -> LayerNorm(mean over L): (B, R, C)   (SwapFusionEncoder.mlp_head)."""
    _need_cuda(x, gamma, beta)
    b, l, r, c = x.shape
    if not x.is_contiguous():
        raise CobevtHipError("mean_layernorm: input must be contiguous")
    out = torch.empty((b, r, c), device=x.device, dtype=x.dtype)
    rc = _L.load().cobevt_layernorm(_p(x), _p(gamma), _p(beta), _p(out), dcode(x.dtype), b * r, c, float(eps), l,
                                    r * c, l * r * c, r, _stream())
    _L.check(rc, "cobevt_layernorm")
    return out


def mean_ln_linear(x, plan):
    """x (B, L, R, C) contiguous bf16 -> plan(mean over L) with plan's folded LayerNorm: (B, R, Cout), one launch
    (SwapFusionEncoder.mlp_head, swap_fusion_modules.py:275-281); None when the shape does not fit the kernel."""
    _need_cuda(x)
    b, l, r, c = x.shape
    if not (USE_GEMM_ROWS3 and x.dtype == torch.bfloat16 and x.is_contiguous() and plan.wfrag_rows is not None and plan.kp_rows == 128
            and plan.K == c and c % 8 == 0 and plan.cout % 8 == 0 and plan.cout <= 4096 and l <= 64 and plan.pre_scale is None
            and plan.stride == 1):
        return None
    out = torch.empty((b, r, plan.cout), device=x.device, dtype=x.dtype)
    dims = (ctypes.c_long * 8)(0, b, l, r, c, plan.cout, int(plan.has_ln), plan.act)

    def cost():
        return 2.0 * b * r * c * plan.cout, float((b * l * r * c + b * r * plan.cout + c * plan.cout) * 2)

    with _timed("gemm_rows|mean%d %d->%d M=%d ln" % (l, c, plan.cout, b * r), cost):
        rc = _L.load().cobevt_mean_linear_rows_small_k(_p(x), _p(plan.wfrag_rows), _p(plan.bias), _p(out), dims,
                                                       ctypes.c_float(plan.ln_eps), _stream())
    _L.check(rc, "cobevt_mean_linear_rows_small_k")
    return out


def tokmap(mode, ncam, hh, ww, w1, w2):
    """Token map tuple for cobevt_window_attention: mode 0 window / 1 grid / 2 stored-partitioned."""
    if hh % w1 or ww % w2:
        raise CobevtHipError("map %dx%d not divisible by window %dx%d" % (hh, ww, w1, w2))
    return (int(mode), int(ncam), int(hh), int(ww), int(w1), int(w2), hh // w1, ww // w2)


def window_attention(q, k, v, out, qmap, kmap, omap, batch, heads, scale, ldq, ldk, ldv, ldo, qoff=0, koff=0, voff=0,
                     ooff=0, bias_table=None, bias_L=1, mask=None, mean_q=False, variant=None, qsplit=None, ksplit=None):
    _need_cuda(q, k, v, out, bias_table, mask)
    code = dcode(q.dtype) | ((ATTN_VARIANT if variant is None else int(variant)) << 8) | \
        ((ATTN_QSPLIT if qsplit is None else int(qsplit)) << 16)
    if k.dtype != q.dtype or v.dtype != q.dtype or out.dtype != q.dtype:
        raise CobevtHipError("window_attention: q/k/v/out dtypes differ")
    L = qmap[6] * qmap[7]
    dims = _ints([code, batch, L, heads, ldq, ldk, ldv, ldo, qoff, koff, voff, ooff,
                  0 if bias_table is None else 1, 0 if bias_table is None else bias_table.shape[0], bias_L,
                  int(mean_q)] + list(qmap) + list(kmap) + list(omap))
    def cost():
        nq = qmap[1] * qmap[4] * qmap[5]
        nk = kmap[1] * kmap[4] * kmap[5]
        esz = q.element_size()
        d = heads * 32
        nbytes = batch * L * (nq + 2 * nk + nq // (qmap[1] if mean_q else 1)) * d * esz
        pairs = nq * nk // (qmap[1] if int(mean_q) == 2 else 1)      # camera-paired: a query copy scores its own camera's keys
        return 4.0 * batch * L * heads * pairs * 32, float(nbytes)

    nq_, nk_ = qmap[1] * qmap[4] * qmap[5], kmap[1] * kmap[4] * kmap[5]
    ks = ATTN_KSPLIT if ksplit is None else int(ksplit)
    # plain bf16 windows of 513 .. 1024 keys (the FAX level-2 / global attention: one whole-map window per agent): K / V resident in LDS,
    # one launch (attention_resident.hip, round 6) instead of the key split + merge of the streaming kernel
    big_resident = (USE_ATTN_BIG_RESIDENT and ksplit is None and q.dtype == torch.bfloat16 and bias_table is None and mask is None and not mean_q
                    and 512 < nk_ <= 1024 and (ATTN_VARIANT if variant is None else int(variant)) == 0 and omap[1] == qmap[1])
    use_ks = (ks > 1 and not big_resident and not mean_q and nk_ >= 1024 and batch * L * heads * ((nq_ + 127) // 128) * ks <= 1024
              and out.is_contiguous() and ldo == out.shape[-1] and ooff == 0 and omap[1] == qmap[1])
    # "fp32_fast": the attention launches go to the third library as well (lib.encoder_scope() is a no-op in every other mode) - fp16
    # queries / probabilities against fp16 (hi, lo) keys / values, csrc/attention.hip kStage16
    with _L.encoder_scope(), _timed("attention|B%d L%d h%d Nq%d Nk%d%s" % (batch, L, heads, nq_, nk_, " ks%d" % ks if use_ks else ""), cost):
        if use_ks:
            rows = out.numel() // out.shape[-1]
            part_out = torch.empty((ks, rows, heads * 32), device=out.device, dtype=out.dtype)
            part_lse = torch.empty((ks, rows, heads), device=out.device, dtype=torch.float32)
            rc = _L.load().cobevt_window_attention_ksplit(_p(q), _p(k), _p(v), _p(out), _p(bias_table), _p(mask), _p(part_out),
                                                          _p(part_lse), dims, ctypes.c_float(scale), ks, rows, _stream())
        else:
            rc = _L.load().cobevt_window_attention(_p(q), _p(k), _p(v), _p(out), _p(bias_table), _p(mask), dims,
                                                   ctypes.c_float(scale), _stream())
    _L.check(rc, "cobevt_window_attention")
    return out


def attention_index_map(tmap, batch, device):
    """int32 (batch, X*Y, ncam*w1*w2): the token -> row map of cobevt_window_attention for one token map (test hook)."""
    L, ntok = tmap[6] * tmap[7], tmap[1] * tmap[4] * tmap[5]
    rows = torch.empty((batch, L, ntok), device=device, dtype=torch.int32)
    _need_cuda(rows)
    rc = _L.load().cobevt_attention_index_map(_ints(tmap), batch, _p(rows), _stream())
    _L.check(rc, "cobevt_attention_index_map")
    return rows


def attention_bias_index(qmap, kmap, bias_L, device):
    """int32 (Nq, Nk): the relative-position table row the attention kernels use for every (query, key) pair (test hook)."""
    nq, nk = qmap[1] * qmap[4] * qmap[5], kmap[1] * kmap[4] * kmap[5]
    idx = torch.empty((nq, nk), device=device, dtype=torch.int32)
    _need_cuda(idx)
    rc = _L.load().cobevt_attention_bias_index(_ints(qmap), _ints(kmap), int(bias_L), _p(idx), _stream())
    _L.check(rc, "cobevt_attention_bias_index")
    return idx


def agent_max(x):
    """(B, L, ...) contiguous -> max over L: (B, ...)   (F-Cooper max-out, fusion_modules/f_cooper_fuse.py:30-36)"""
    _need_cuda(x)
    if not x.is_contiguous():
        raise CobevtHipError("agent_max: input must be contiguous")
    b, l = x.shape[:2]
    out = torch.empty((b,) + tuple(x.shape[2:]), device=x.device, dtype=x.dtype)
    per = out[0].numel()
    rc = _L.load().cobevt_agent_max(_p(x), _p(out), dcode(x.dtype), b, l, per, _stream())
    _L.check(rc, "cobevt_agent_max")
    return out


def ray_embed(i_inv, e_inv, image_plane, w_img, w_cam, hw, dim, dtype):
    _need_cuda(i_inv, e_inv, image_plane, w_img, w_cam)
    bn = i_inv.shape[0]
    out = torch.empty((bn, hw, dim), device=i_inv.device, dtype=dtype)
    rc = _L.load().cobevt_fax_ray_embed(_p(i_inv), _p(e_inv), _p(image_plane), _p(w_img), _p(w_cam), _p(out),
                                        dcode(dtype), bn, hw, dim, _stream())
    _L.check(rc, "cobevt_fax_ray_embed")
    return out


def batch_broadcast(x):
    """True for expand()-ed views whose leading dimension has stride 0 over one contiguous slice."""
    return x.dim() >= 2 and x.shape[0] > 1 and x.stride(0) == 0 and x[0].is_contiguous()


def bev_embed(e_inv, world, w_bev, b_bev, w_cam, x, n):
    """x: (B, HW, D), or a batch-broadcast view (stride 0 over B) of one (HW, D) prior -> query (B, n, HW, D)"""
    _need_cuda(e_inv, world, w_bev, b_bev, w_cam, x)
    b, hw, d = x.shape
    bcast = batch_broadcast(x)
    if not bcast:
        x = x.contiguous()
    out = torch.empty((b, n, hw, d), device=x.device, dtype=x.dtype)
    rc = _L.load().cobevt_fax_bev_embed(_p(e_inv), _p(world), _p(w_bev), _p(b_bev), _p(w_cam), _p(x), _p(out),
                                        dcode(x.dtype), b, n, hw, d, int(bcast), _stream())
    _L.check(rc, "cobevt_fax_bev_embed")
    return out


def bev_embed_linear(e_inv, world, w_bev, b_bev, w_cam, x, n, plan):
    """plan(bev_embed(...)) : (B, HW, D) -> (B, n, HW, N) without materialising the (B, n, HW, D) query (the dense-row GEMM
    produces its A rows on the fly) when plan is a LayerNorm-folded Linear whose row fits one K-tile; otherwise the two
    launches."""
    b, hw, d = x.shape
    bcast = batch_broadcast(x)
    wave_level = d == 128 and plan.cout == 128 and b * n <= 32           # bev_query.hip's shape (else the call lands in gemm_rows3.hip)
    fused3 = ((USE_EMBED_GEMM3 == 2 or (USE_EMBED_GEMM3 and wave_level)) and USE_GEMM_ROWS3 and plan.code == BF16 and plan.has_ln
              and plan.K == d and d <= 128 and d % 8 == 0
              and plan.kp_rows == 128 and plan.wfrag_rows is not None and hw % 32 == 0 and plan.stride == 1
              and plan.pre_scale is None and plan.act == 0 and plan.cout % 8 == 0 and (bcast or x.is_contiguous()))
    if fused3:
        _need_cuda(e_inv, world, w_bev, b_bev, w_cam, x)
        out = torch.empty((b, n, hw, plan.cout), device=x.device, dtype=x.dtype)
        dims = (ctypes.c_long * 8)(plan.code, b, n, hw, d, plan.cout, 1, int(bcast))

        def cost3():
            m = b * n * hw
            return 2.0 * m * plan.cout * d, float((hw if bcast else x.numel() // d) * d * 2 + plan.cout * d * 2 + out.numel() * 2)

        with _timed("gemm_rows|embed %d->%d M=%d ln r32" % (d, plan.cout, b * n * hw), cost3):
            rc = _L.load().cobevt_bev_embed_linear_rows_small_k(_p(e_inv), _p(world), _p(w_bev), _p(b_bev), _p(w_cam), _p(x),
                                                                _p(plan.wfrag_rows), _p(plan.bias), _p(out), dims,
                                                                ctypes.c_float(plan.ln_eps), _stream())
        _L.check(rc, "cobevt_bev_embed_linear_rows_small_k")
        return out
    fused = (USE_EMBED_GEMM and USE_GEMM_ROWS and plan.has_ln and ln_fusable(plan) and plan.K == d and hw % 128 == 0
             and plan.stride == 1 and plan.pre_scale is None and plan.act == 0 and x.is_contiguous())
    if not fused:
        return linear(bev_embed(e_inv, world, w_bev, b_bev, w_cam, x, n), plan)
    _need_cuda(e_inv, world, w_bev, b_bev, w_cam, x)
    out = torch.empty((b, n, hw, plan.cout), device=x.device, dtype=x.dtype)
    dims = (ctypes.c_long * 8)(plan.code, b, n, hw, d, plan.cout, plan.kp_rows, 1)

    def cost():
        m = b * n * hw
        esz = 2 if plan.code == BF16 else 4
        return 2.0 * m * plan.cout * d, float(x.numel() * esz + plan.cout * d * esz + out.numel() * esz)

    with _timed("gemm_rows|embed %d->%d M=%d ln" % (d, plan.cout, b * n * hw), cost):
        rc = _L.load().cobevt_bev_embed_linear_rows(_p(e_inv), _p(world), _p(w_bev), _p(b_bev), _p(w_cam), _p(x),
                                                    _p(plan.wgt_rows), _p(plan.bias), _p(out), dims,
                                                    ctypes.c_float(plan.ln_eps), _stream())
    _L.check(rc, "cobevt_bev_embed_linear_rows")
    return out


def stem_pool(x, plan):
    """maxpool3x3s2(conv2d(x, plan)) for the ResNet stem plan (7x7 / s2 / pad 3 on the fp32 image, 64 couts, ReLU):
    one launch when the image sides are multiples of 4, else the two kernels."""
    _need_cuda(x)
    n, h, w, cin = x.shape
    fused = (USE_STEM_POOL and USE_STEM and plan.wgt_stem is not None and plan.cout == 64 and plan.act == 1
             and h % 4 == 0 and w % 4 == 0 and x.dtype == torch.float32 and x.is_contiguous())
    if not fused:
        return maxpool3x3s2(conv2d(x, plan))
    out = torch.empty((n, h // 4, w // 4, 64), device=x.device, dtype=plan.dtype)
    dims = _ints([plan.code, n, h, w])

    def cost():
        esz = 2 if plan.code == BF16 else 4
        return 2.0 * n * (h // 2) * (w // 2) * 64 * 147, float(x.numel() * 4 + out.numel() * esz)

    with _timed("stem7x7|pool %dx%dx%d" % (n, h, w), cost):
        rc = _L.load().cobevt_stem_conv7x7s2_pool(_p(x), _p(plan.wgt_stem), _p(plan.bias), _p(out), dims, _stream())
    _L.check(rc, "cobevt_stem_conv7x7s2_pool")
    return out


def stem_pool_u8(x, lut, plan):
    """stem_pool on uint8 camera frames (N, H, W, 3) + the (3, 256) fp32 normalisation table (host/rgb_preprocessor.py): the
    reference's host-side normalisation and the fp32 image upload folded into the stem's gather.  No fallback: the frame sides must
    be multiples of 4 (every OPV2V / nuScenes resolution is)."""
    _need_cuda(x, lut)
    n, h, w, cin = x.shape
    if not (plan.wgt_stem is not None and plan.cout == 64 and plan.act == 1 and h % 4 == 0 and w % 4 == 0 and cin == 3
            and x.dtype == torch.uint8 and x.is_contiguous() and lut.dtype == torch.float32 and tuple(lut.shape) == (3, 256)
            and lut.is_contiguous()):
        raise CobevtHipError("stem_pool_u8: needs contiguous uint8 frames (N, H, W, 3) with H, W multiples of 4, a (3, 256) fp32 table "
                             "and the ResNet stem plan; got %s %s" % (tuple(x.shape), x.dtype))
    out = torch.empty((n, h // 4, w // 4, 64), device=x.device, dtype=plan.dtype)
    dims = _ints([plan.code, n, h, w])

    def cost():
        esz = 2 if plan.code == BF16 else 4
        return 2.0 * n * (h // 2) * (w // 2) * 64 * 147, float(x.numel() + out.numel() * esz)

    with _timed("stem7x7|pool u8 %dx%dx%d" % (n, h, w), cost):
        rc = _L.load().cobevt_stem_conv7x7s2_pool_u8(_p(x), _p(lut), _p(plan.wgt_stem), _p(plan.bias), _p(out), dims, _stream())
    _L.check(rc, "cobevt_stem_conv7x7s2_pool_u8")
    return out


def host_fetch(src_pinned, dst, blocks=0):
    """dst (device) <- src_pinned (pinned host tensor of the same shape / dtype) by a kernel that reads the host buffer over PCIe
    (cobevt_host_fetch): capturable in a HIP graph, runs beside the compute kernels."""
    _need_cuda(dst)
    if src_pinned.is_cuda or not src_pinned.is_pinned():
        raise CobevtHipError("host_fetch: the source must be a pinned host tensor (tensor.pin_memory())")
    nbytes = dst.numel() * dst.element_size()
    if (tuple(src_pinned.shape) != tuple(dst.shape) or src_pinned.dtype != dst.dtype or not src_pinned.is_contiguous() or not dst.is_contiguous()
            or nbytes % 16):
        raise CobevtHipError("host_fetch: contiguous tensors of one shape / dtype, a multiple of 16 bytes")
    rc = _L.load().cobevt_host_fetch(ctypes.c_void_p(src_pinned.data_ptr()), _p(dst), nbytes, int(blocks), _stream())
    _L.check(rc, "cobevt_host_fetch")
    return dst


def maxpool3x3s2(x):
    _need_cuda(x)
    n, h, w, c = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    out = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    rc = _L.load().cobevt_maxpool3x3s2(_p(x), _p(out), dcode(x.dtype), n, h, w, c, _stream())
    _L.check(rc, "cobevt_maxpool3x3s2")
    return out


def to_nhwc(x, dtype):
    """Any (N,C,H,W)-shaped CUDA tensor (arbitrary strides, bf16/fp32) -> contiguous (N,H,W,C) in `dtype`.
    Zero-copy when x already is a channels-last view of the right dtype."""
    _need_cuda(x)
    n, c, h, w = x.shape
    v = x.permute(0, 2, 3, 1)
    if x.dtype == dtype and v.is_contiguous():
        return v
    out = torch.empty((n, h, w, c), device=x.device, dtype=dtype)
    strides = (ctypes.c_long * 4)(*x.stride())
    rc = _L.load().cobevt_to_nhwc(_p(x), dcode(x.dtype), _p(out), dcode(dtype), n, c, h, w, strides, _stream())
    _L.check(rc, "cobevt_to_nhwc")
    return out


def from_nhwc(x, dtype):
    """Contiguous (N,H,W,C) -> contiguous (N,C,H,W) in `dtype`."""
    _need_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=dtype)
    strides = (ctypes.c_long * 4)(*out.stride())
    rc = _L.load().cobevt_from_nhwc(_p(x), dcode(x.dtype), _p(out), dcode(dtype), n, c, h, w, strides, _stream())
    _L.check(rc, "cobevt_from_nhwc")
    return out


def copy_into_interior(x, dst):
    """dst[:, :h, :w, :] = x for contiguous (N, h, w, C) `x` and a contiguous, larger (N, H, W, C) `dst` of the same dtype - the
    zero-padded key map of a FAX level without image features (fax_modules.py:392-396).  cobevt_from_nhwc with the interior's
    strides: no torch arithmetic on the inference path."""
    _need_cuda(x, dst)
    n, h, w, c = x.shape
    if dst.dtype != x.dtype or not x.is_contiguous() or not dst.is_contiguous() or dst.shape[0] != n or dst.shape[3] != c \
            or dst.shape[1] < h or dst.shape[2] < w:
        raise CobevtHipError("copy_into_interior: dst must be a contiguous (N, >= h, >= w, C) map of x's dtype")
    sN, sH, sW, sC = dst.stride()
    strides = (ctypes.c_long * 4)(sN, sC, sH, sW)                     # (N, C, H, W) order of the C entry point
    rc = _L.load().cobevt_from_nhwc(_p(x), dcode(x.dtype), _p(dst), dcode(dst.dtype), n, c, h, w, strides, _stream())
    _L.check(rc, "cobevt_from_nhwc")
    return dst


def regroup(x, record_len, max_cav):
    """x: (N, ...) contiguous, record_len int32 device (B,) -> (B, max_cav, ...), mask (B, max_cav) fp32."""
    _need_cuda(x, record_len)
    if record_len.dtype != torch.int32 or not x.is_contiguous():
        raise CobevtHipError("regroup: record_len must be int32 on the device and x contiguous")
    b = record_len.shape[0]
    per = x[0].numel()
    out = torch.empty((b, max_cav) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    mask = torch.empty((b, max_cav), device=x.device, dtype=torch.float32)
    rc = _L.load().cobevt_regroup(_p(x), _p(record_len), _p(out), _p(mask), dcode(x.dtype), b, max_cav, per, _stream())
    _L.check(rc, "cobevt_regroup")
    return out, mask


def sttf_warp(x, tmat, cav_mask, discrete_ratio, downsample_rate, want_mask=True, record_len=None, max_cav=None):
    """x: (B, L, H, W, C) contiguous ; tmat (B, L, 4, 4) fp32 -> warped (B,L,H,W,C), com_mask (B,H,W,1,L)|None.
    With record_len (int32 device (B,)) x is the un-grouped agent batch (N, H, W, C): regroup + warp in one launch,
    returning (warped, com_mask, cav_mask (B, max_cav))."""
    _need_cuda(x, tmat, cav_mask, record_len)
    if tmat.dtype != torch.float32 or not tmat.is_contiguous() or not x.is_contiguous():
        raise CobevtHipError("sttf_warp: tmat must be contiguous fp32, x contiguous")
    if record_len is not None:
        if record_len.dtype != torch.int32 or x.dim() != 4:
            raise CobevtHipError("sttf_warp: record_len must be int32 on the device and x (N, H, W, C)")
        b, l = record_len.shape[0], int(max_cav)
        _, h, w, c = x.shape
        out = torch.empty((b, l, h, w, c), device=x.device, dtype=x.dtype)
        cav = torch.empty((b, l), device=x.device, dtype=torch.float32)
    else:
        b, l, h, w, c = x.shape
        out = torch.empty_like(x)
        cav = None
    if tuple(tmat.shape[:2]) != (b, l):
        raise CobevtHipError("sttf_warp: tmat must be (B, L, 4, 4)")
    com = torch.empty((b, h, w, 1, l), device=x.device, dtype=torch.float32) if want_mask else None
    rc = _L.load().cobevt_sttf_warp(_p(x), _p(tmat), _p(cav_mask), _p(out), _p(com), _p(record_len), _p(cav),
                                    dcode(x.dtype), b, l, h, w, c,
                                    ctypes.c_float(discrete_ratio), ctypes.c_float(downsample_rate), _stream())
    _L.check(rc, "cobevt_sttf_warp")
    return (out, com, cav) if record_len is not None else (out, com)


def pairwise_warp(x, pairwise, record_len, max_cav, discrete_ratio, downsample_rate):
    """x (sum(record_len), H, W, C) un-grouped agent maps; pairwise (B, L, L, 4, 4) fp32; record_len device int32 (B,) ->
    nb (B, L, L, H, W, C): nb[b, i, j] = agent j in agent i's frame; roi (B, L, L, H, W) fp32 (v2v_fuse.py:59-103)."""
    _need_cuda(x, pairwise, record_len)
    if record_len.dtype != torch.int32 or x.dim() != 4 or pairwise.dtype != torch.float32:
        raise CobevtHipError("pairwise_warp: record_len int32 on the device, x (N, H, W, C), pairwise fp32")
    b, l = record_len.shape[0], int(max_cav)
    _, h, w, c = x.shape
    if tuple(pairwise.shape) != (b, l, l, 4, 4):
        raise CobevtHipError("pairwise_warp: pairwise must be (B, max_cav, max_cav, 4, 4)")
    nb = torch.empty((b, l, l, h, w, c), device=x.device, dtype=x.dtype)
    roi = torch.empty((b, l, l, h, w), device=x.device, dtype=torch.float32)
    rc = _L.load().cobevt_pairwise_warp(_p(x), _p(pairwise.contiguous()), _p(record_len), _p(nb), _p(roi), dcode(x.dtype), b, l, h, w, c,
                                        ctypes.c_float(discrete_ratio), ctypes.c_float(downsample_rate), _stream())
    _L.check(rc, "cobevt_pairwise_warp")
    return nb, roi


def agent_message_reduce(msg, ego, roi, record_len, mode):
    """msg (B, L, L, H, W, C), ego (N, H, W, C), roi (B, L, L, H, W) -> (N, H, W, C): mean ('avg') | max over the valid source agents of
    (msg + ego) * roi  (v2v_fuse.py:108-119)"""
    _need_cuda(msg, ego, roi, record_len)
    b, l, _, h, w, c = msg.shape
    out = torch.zeros_like(ego)
    rc = _L.load().cobevt_agent_message_reduce(_p(msg), _p(ego), _p(roi), _p(record_len), _p(out), dcode(msg.dtype), b, l, h * w, c,
                                               {"avg": 0, "max": 1}[mode], _stream())
    _L.check(rc, "cobevt_agent_message_reduce")
    return out


def gru_zero_state(x):
    """(..., 2C) [update | candidate] pre-activations -> (..., C): sigmoid(update) * tanh(candidate)"""
    _need_cuda(x)
    c = x.shape[-1] // 2
    out = torch.empty(x.shape[:-1] + (c,), device=x.device, dtype=x.dtype)
    rc = _L.load().cobevt_gru_zero_state(_p(x), _p(out), dcode(x.dtype), x.numel() // (2 * c), c, _stream())
    _L.check(rc, "cobevt_gru_zero_state")
    return out


def agent_softmax_sum(score, nb, roi, record_len, n_agents, use_mask=True):
    """score (B*L*L*H*W, lds) (column 0), nb (B, L, L, H, W, C), roi -> (n_agents, H, W, C)  (disconet_fuse.py:141-150)"""
    _need_cuda(score, nb, roi, record_len)
    b, l, _, h, w, c = nb.shape
    out = torch.zeros((n_agents, h, w, c), device=nb.device, dtype=nb.dtype)
    rc = _L.load().cobevt_agent_softmax_sum(_p(score), score.shape[-1], _p(nb), _p(roi), _p(record_len), _p(out), dcode(nb.dtype),
                                            b, l, h * w, c, int(bool(use_mask)), _stream())
    _L.check(rc, "cobevt_agent_softmax_sum")
    return out


def invert_small(m):
    """Batched inverse of (..., 3, 3) or (..., 4, 4) fp32 matrices on the device."""
    _need_cuda(m)
    d = m.shape[-1]
    x = m.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    rc = _L.load().cobevt_invert_small(_p(x), _p(out), x.numel() // (d * d), d, _stream())
    _L.check(rc, "cobevt_invert_small")
    return out


def resize_nhwc(x, ho, wo, mode):
    """(N,H,W,C) -> (N,ho,wo,C); mode 'nearest' (F.interpolate default) or 'bilinear' (align_corners=True)."""
    _need_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    rc = _L.load().cobevt_resize_nhwc(_p(x), _p(out), dcode(x.dtype), n, h, w, c, ho, wo, 0 if mode == "nearest" else 1,
                                      _stream())
    _L.check(rc, "cobevt_resize_nhwc")
    return out


def channel_affine(x, scale, shift):
    """x (N, C, ...) contiguous fp32 -> x * scale[c] + shift[c]"""
    _need_cuda(x, scale, shift)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    n, c = x.shape[0], x.shape[1]
    rc = _L.load().cobevt_channel_affine(_p(x), _p(scale), _p(shift), _p(out), n, c, x.numel() // (n * c), _stream())
    _L.check(rc, "cobevt_channel_affine")
    return out


def chain_next_fusable(plan_n, c):
    """Can `plan_n` (a Linear / 1x1 conv plan reading C-channel rows) ride at the end of the fused row chain?"""
    return (plan_n is not None and plan_n.wfrag_rows is not None and plan_n.stride == 1 and plan_n.kp_rows == 128 and plan_n.K == c
            and plan_n.cout % 8 == 0 and plan_n.cout <= 768 and plan_n.pre_scale is None and not plan_n.pre_relu
            and plan_n.act <= 2)


def attn_mlp_chain(a, skip, plan_p, plan_1, plan_2, post_ln=None, next_plan=None):
    """out = postLN( y + fc2(GELU(fc1(LN(y)))) ),  y = proj(a) + skip.   a, skip: (..., C) contiguous.
    plan_1 must be built with ln=<prenorm> (affine folded) and act=GELU; post_ln = (gamma, beta, eps) fp32 or None.
    One fused launch in bf16 mode when the shapes fit (C <= 128, hidden <= 256); otherwise three GEMM launches.
    next_plan: the Linear / 1x1-conv plan that consumes `out` next (LayerNorm folded via ln=..., BN folded, act); when
    given the call returns (out, next_plan(out)) - computed inside the same launch when fused."""
    _need_cuda(a, skip)
    c, hd = plan_p.cout, plan_1.cout
    skip_rows = 0
    if skip is not None and batch_broadcast(skip) and skip.shape == a.shape:
        skip_rows = skip[0].numel() // c           # one slice shared by the whole batch: the kernel indexes it modulo
    # fp32 storage (round 6, csrc/row_chain_f32.hip): the C = 128 / hidden 256 chain of every FAX level and of the camera fusion stage
    f32 = a.dtype == torch.float32
    fusable = (USE_ROW_CHAIN and (a.dtype == torch.bfloat16 or (f32 and USE_ROW_CHAIN_F32 and c == 128 and hd == 256 and plan_2.kp_rows == 256))
               and plan_p.wfrag_rows is not None and plan_1.wfrag_rows is not None
               and plan_2.wfrag_rows is not None and plan_2.kp_rows <= 256 and plan_1.has_ln and plan_1.act == 2 and plan_p.act == 0
               and plan_2.act == 0 and not plan_p.has_ln and not plan_2.has_ln and plan_p.K == c and plan_1.K == c
               and plan_2.K == hd and plan_2.cout == c and c <= 128 and c % 8 == 0 and hd <= 256 and hd % 8 == 0
               and plan_p.kp_rows == 128 and plan_1.kp_rows == 128 and a.shape[-1] == c and a.is_contiguous()
               and (skip is None or skip.dtype == a.dtype)
               and (skip is None or skip_rows or (skip.is_contiguous() and skip.shape == a.shape)))
    if not fusable:
        y = linear(a, plan_p, residual=skip.contiguous() if skip is not None else None)
        z = linear(linear(y, plan_1), plan_2, residual=y)
        z = layernorm(z, post_ln[0], post_ln[1], post_ln[2]) if post_ln is not None else z
        return (z, linear(z, next_plan)) if next_plan is not None else z
    m = a.numel() // c
    out = torch.empty_like(a)
    fuse_next = USE_CHAIN_NEXT and chain_next_fusable(next_plan, c)
    nn_ = next_plan.cout if fuse_next else 0
    out_next = torch.empty(a.shape[:-1] + (nn_,), device=a.device, dtype=a.dtype) if fuse_next else None
    dims = _ints([plan_p.code, m, c, hd, plan_2.kp_rows, nn_, int(next_plan.has_ln) if fuse_next else 0,
                  next_plan.act if fuse_next else 0, ROW_CHAIN_ROWS, skip_rows])
    pg, pb, pe = post_ln if post_ln is not None else (None, None, 0.0)

    def cost():
        flops = 2.0 * m * (c * c + 2 * c * hd + c * nn_)
        esz = a.element_size()
        return flops, float((2 * m + (0 if skip is None else skip_rows or m)) * c * esz + m * nn_ * esz + (c * c + 2 * c * hd + c * nn_) * esz)

    with _timed("row_chain|C%d H%d M=%d%s%s" % (c, hd, m, " post" if post_ln is not None else "",
                                                  " +next%d" % nn_ if fuse_next else ""), cost):
        rc = _L.load().cobevt_attn_mlp_chain(_p(a), _p(skip), _p(out), _p(plan_p.wfrag_rows), _p(plan_p.bias),
                                             _p(plan_1.wfrag_rows), _p(plan_1.bias), _p(plan_2.wfrag_rows), _p(plan_2.bias),
                                             _p(pg), _p(pb), _p(next_plan.wfrag_rows) if fuse_next else None,
                                             _p(next_plan.bias) if fuse_next else None, _p(out_next), dims,
                                             ctypes.c_float(plan_1.ln_eps), ctypes.c_float(pe),
                                             ctypes.c_float(next_plan.ln_eps if fuse_next else 0.0), _stream())
    _L.check(rc, "cobevt_attn_mlp_chain")
    if next_plan is None:
        return out
    return out, (out_next if fuse_next else linear(out, next_plan))


def proj_chain_fusable(x, plan_p, next_plan, residual):
    c = x.shape[-1]
    return (USE_PROJ_CHAIN and x.dtype == torch.bfloat16 and c == 128 and x.is_contiguous() and plan_p.wfrag_rows is not None
            and plan_p.kp_rows == 128 and plan_p.K == c and plan_p.cout == c and plan_p.act == 0 and not plan_p.has_ln
            and plan_p.stride == 1 and chain_next_fusable(next_plan, c)
            and (residual is None or (residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype)))


def proj_chain(x, plan_p, next_plan, residual=None, out_next=None):
    """next_plan(plan_p(x) + residual) on (..., 128) rows in one launch (cobevt_proj_chain); plan_p may carry the pre-activation
    BatchNorm -> ReLU of its input, next_plan a folded LayerNorm.  The intermediate map is not materialised."""
    _need_cuda(x, residual)
    c, nn_ = 128, next_plan.cout
    m = x.numel() // c
    if out_next is None:
        out_next = torch.empty(x.shape[:-1] + (nn_,), device=x.device, dtype=x.dtype)
    elif out_next.numel() != m * nn_ or not out_next.is_contiguous() or out_next.dtype != x.dtype:
        raise CobevtHipError("proj_chain: `out_next` must be a contiguous (.., %d) buffer of dtype %s" % (nn_, x.dtype))
    dims = _ints([0, m, c, nn_, int(next_plan.has_ln), next_plan.act, 0, plan_p.pre_relu, 0 if USE_PROJ_CHAIN_WAVE else 1])

    def cost():
        return 2.0 * m * (c * c + c * nn_), float(m * (c * (2 if residual is not None else 1) + nn_) * 2 + (c * c + c * nn_) * 2)

    with _timed("row_chain|proj C%d M=%d +next%d" % (c, m, nn_), cost):
        rc = _L.load().cobevt_proj_chain(_p(x), _p(plan_p.pre_scale), _p(plan_p.pre_shift), _p(residual), _p(plan_p.wfrag_rows),
                                         _p(plan_p.bias), None, _p(next_plan.wfrag_rows), _p(next_plan.bias), _p(out_next), dims,
                                         ctypes.c_float(next_plan.ln_eps), _stream())
    _L.check(rc, "cobevt_proj_chain")
    return out_next


def proj_chain_kv_fusable(x, plan_key, plan_val, next_key, next_val, residual):
    k = x.shape[-1]
    if not (USE_PROJ_CHAIN_KV and x.dtype == torch.bfloat16 and x.is_contiguous() and k in (256, 384, 512)):
        return False
    for pl in (plan_key, plan_val):
        if pl is None:
            continue
        if not (pl.wfrag_rows is not None and pl.kp_rows == k and pl.K == k and pl.cout == 128 and pl.act == 0 and not pl.has_ln
                and pl.stride == 1):
            return False
    for pn in (next_key, next_val):
        if not (chain_next_fusable(pn, 128) and pn.act == 0):
            return False
    return (plan_val is not None and next_key.cout == next_val.cout and next_key.has_ln == next_val.has_ln and next_key.ln_eps == next_val.ln_eps
            and (residual is None or (plan_key is not None and residual.is_contiguous() and residual.dtype == x.dtype
                                      and tuple(residual.shape) == tuple(x.shape[:-1]) + (128,))))


def proj_chain_kv(x, plan_key, plan_val, next_key, next_val, residual=None, out_k=None, out_v=None):
    """(next_key(plan_key(x) + residual), next_val(plan_val(x))) on (..., K) feature rows, K = 256 / 384 / 512 -> two (..., Nn) maps, ONE
    launch (cobevt_proj_chain_kv); neither 128-channel intermediate map reaches HBM.  plan_key None (`no_image_features`: the key is
    the ray embedding alone, fax_modules.py:392-396) is not served here."""
    _need_cuda(x, residual)
    k, nn_ = x.shape[-1], next_key.cout
    m = x.numel() // k
    outs = []
    for o in (out_k, out_v):
        if o is None:
            o = torch.empty(x.shape[:-1] + (nn_,), device=x.device, dtype=x.dtype)
        elif o.numel() != m * nn_ or not o.is_contiguous() or o.dtype != x.dtype:
            raise CobevtHipError("proj_chain_kv: result buffers must be contiguous (.., %d) of dtype %s" % (nn_, x.dtype))
        outs.append(o)
    ptrs = (ctypes.c_void_p * 18)()
    for s_, (pp, pn, res, o) in enumerate(((plan_key, next_key, residual, outs[0]), (plan_val, next_val, None, outs[1]))):
        vals = (pp.pre_scale, pp.pre_shift, pp.wfrag_rows, pp.bias, res, pn.wfrag_rows, pn.bias, None, o)
        for j, t in enumerate(vals):
            ptrs[9 * s_ + j] = None if t is None else t.data_ptr()
    dims = _ints([0, m, k, nn_, int(next_key.has_ln), 2, plan_key.pre_relu, 0, plan_val.pre_relu, 0])

    def cost():
        return 2.0 * 2 * m * (k * 128 + 128 * nn_), float(m * (k + (128 if residual is not None else 0) + 2 * nn_) * 2 + 2 * (k * 128 + 128 * nn_) * 2)

    with _timed("row_chain|proj kv K%d M=%d +next%d" % (k, m, nn_), cost):
        rc = _L.load().cobevt_proj_chain_kv(_p(x), ptrs, dims, ctypes.c_float(next_key.ln_eps), _stream())
    _L.check(rc, "cobevt_proj_chain_kv")
    return outs[0], outs[1]


def swap_stage_fusable(qkv, x, tmap, heads, plan_p, plan_1, plan_2, next_plan, mask):
    """does one SwapFusionBlock half fit the single-launch kernel (cobevt_swap_fusion_stage)?"""
    c = x.shape[-1]
    nk = tmap[1] * tmap[4] * tmap[5]
    return (USE_SWAP_STAGE and x.dtype == torch.bfloat16 and qkv.dtype == torch.bfloat16 and c == 128 and heads == 4
            and tmap[0] in (0, 1) and nk <= 384 and x.is_contiguous() and qkv.is_contiguous() and qkv.shape[-1] == 3 * c
            and plan_p.wfrag_rows is not None and plan_1.wfrag_rows is not None and plan_2.wfrag_rows is not None
            and plan_p.kp_rows == 128 and plan_1.kp_rows == 128 and plan_2.kp_rows in (128, 256) and plan_1.has_ln
            and plan_1.act == 2 and plan_p.act == 0 and plan_2.act == 0 and not plan_p.has_ln and not plan_2.has_ln
            and plan_p.K == c and plan_1.K == c and plan_2.K == plan_1.cout and plan_2.cout == c and plan_1.cout % 8 == 0
            and plan_1.cout <= 256 and plan_p.cout == c
            and (next_plan is None or (chain_next_fusable(next_plan, c) and next_plan.cout <= 384 and next_plan.has_ln
                                       and next_plan.act == 0))
            and (mask is None or (mask.dtype == torch.float32 and mask.is_contiguous())))


SWAP_STAGE_BIAS_FLOATS = 10240     # the kernel copies the relative-position table as a fixed 40-KB image (csrc/swap_stage.hip)


def swap_stage(qkv, x, tmap, batch, heads, scale, bias_table, bias_L, mask, plan_p, plan_1, plan_2, next_plan=None):
    """One SwapFusionBlock half: x (b, l, h, w, 128) + its qkv = to_qkv(LN(x)) (b, l, h, w, 384) -> x_out (and the next
    half's to_qkv(LN(x_out)) when next_plan is given).  bias_table: the flat SWAP_STAGE_BIAS_FLOATS image of
    swap_fusion_modules.Attention.stage_bias_table().  Caller checks swap_stage_fusable() first."""
    _need_cuda(qkv, x, bias_table, mask)
    c, hd = 128, plan_1.cout
    out = torch.empty_like(x)
    nn_ = next_plan.cout if next_plan is not None else 0
    qkv_next = torch.empty(x.shape[:-1] + (nn_,), device=x.device, dtype=x.dtype) if next_plan is not None else None
    bias_rows = (2 * bias_L - 1) * (2 * tmap[4] - 1) * (2 * tmap[5] - 1)
    if bias_table.numel() != SWAP_STAGE_BIAS_FLOATS or bias_table.dtype != torch.float32 or not bias_table.is_contiguous():
        raise CobevtHipError("swap_stage: bias table must be the flat fp32 image of stage_bias_table()")
    dims = _ints([0, batch, c, heads, hd, plan_2.kp_rows, nn_, bias_rows, bias_L])
    m = x.numel() // c
    nk = tmap[1] * tmap[4] * tmap[5]

    def cost():
        flops = 4.0 * m * nk * c + 2.0 * m * (c * c + 2 * c * hd + c * nn_)
        return flops, float(m * (3 * c + 2 * c + nn_) * 2 + (c * c + 2 * c * hd + c * nn_) * 2)

    with _timed("swap_stage|mode%d M=%d Nk%d H%d%s" % (tmap[0], m, nk, hd, " +next%d" % nn_ if nn_ else ""), cost):
        rc = _L.load().cobevt_swap_fusion_stage(
            _p(qkv), _p(x), _p(out), _p(qkv_next), _ints(tmap), _p(bias_table), _p(mask), _p(plan_p.wfrag_rows), _p(plan_p.bias),
            _p(plan_1.wfrag_rows), _p(plan_1.bias), _p(plan_2.wfrag_rows), _p(plan_2.bias),
            _p(next_plan.wfrag_rows) if next_plan is not None else None, _p(next_plan.bias) if next_plan is not None else None,
            dims, ctypes.c_float(scale), ctypes.c_float(plan_1.ln_eps), ctypes.c_float(next_plan.ln_eps if next_plan is not None else 0.0),
            _stream())
    _L.check(rc, "cobevt_swap_fusion_stage")
    return out, qkv_next


# ----------------------------------------------------------------------------------------------
# downstream of the hot path: logits -> probabilities / class maps -> per-class pixel counts
# ----------------------------------------------------------------------------------------------
def softmax_argmax(seg_logits):
    """(N, C, H, W) logits -> (softmax(dim=1) fp32 (N, C, H, W), argmax of the probabilities int64 (N, H, W));
    CameraBevPostprocessor.softmax_argmax, camera_bev_postprocessor.py:55-59."""
    _need_cuda(seg_logits)
    if seg_logits.dim() != 4:
        raise CobevtHipError("softmax_argmax expects (N, C, H, W) logits, got %s" % (tuple(seg_logits.shape),))
    x = seg_logits.contiguous()
    n, c, h, w = x.shape
    prob = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    seg_map = torch.empty((n, h, w), device=x.device, dtype=torch.int64)
    rc = _L.load().cobevt_softmax_argmax(_p(x), _p(prob), _p(seg_map), dcode(x.dtype), n, c, h * w, _stream())
    _L.check(rc, "cobevt_softmax_argmax")
    return prob, seg_map


def seg_class_counts(pred, gt, num_classes):
    """pred, gt: (N, H, W) integer label maps -> (N, num_classes, 3) int64 counts (n_ii, t_i, n_ij) per class - what
    seg_utils.py:25-50 (mean_IU) and :6-21 (mean_precision) reduce their masks to.  Labels outside [0, num_classes)
    raise (the reference would treat them as further classes)."""
    _need_cuda(pred, gt)
    if pred.shape != gt.shape or pred.dim() != 3:
        raise CobevtHipError("seg_class_counts expects two (N, H, W) maps, got %s and %s" % (tuple(pred.shape), tuple(gt.shape)))
    p64, g64 = pred.to(torch.int64).contiguous(), gt.to(torch.int64).contiguous()
    n, h, w = p64.shape
    counts = torch.empty((n, num_classes + 1, 3), device=pred.device, dtype=torch.int64)
    rc = _L.load().cobevt_seg_class_counts(_p(p64), _p(g64), _p(counts), n, h * w, num_classes, _stream())
    _L.check(rc, "cobevt_seg_class_counts")
    host = counts.cpu()
    if int(host[:, num_classes, 0].sum()) or int(host[:, num_classes, 1].sum()):
        raise CobevtHipError("seg_class_counts: labels outside [0, %d)" % num_classes)
    return host[:, :num_classes]


# ----------------------------------------------------------------------------------------------
# MBConv pieces of the nuScenes EfficientNet backbone
# ----------------------------------------------------------------------------------------------
class DepthwisePlan(object):
    """k x k depthwise conv (groups == channels) + folded BatchNorm + activation, TensorFlow-"same" static padding:
    `pad` = (before, after) zero rows / columns, the kernel takes the before part and bounds-checks the rest."""

    def __init__(self, weight, bn=None, stride=1, pad=(0, 0), act=0, dtype=torch.bfloat16, device="cuda"):
        w = weight.detach().double().cpu()                      # (C, 1, k, k)
        c, one, kh, kw = w.shape
        if one != 1 or kh != kw:
            raise CobevtHipError("DepthwisePlan expects a (C, 1, k, k) weight")
        b = torch.zeros(c, dtype=torch.float64)
        if bn is not None:
            s_, sh = bn_affine(bn)
            w = w * s_.cpu()[:, None, None, None]
            b = sh.cpu().double()
        if c % 8 != 0:
            raise CobevtHipError("depthwise conv needs a multiple of 8 channels, got %d" % c)
        self.c, self.k, self.stride, self.act = c, kh, int(stride), int(act)
        self.pad0, self.pad1 = int(pad[0]), int(pad[1])
        self.dtype, self.code = dtype, dcode(dtype)
        self.wgt = w[:, 0].permute(1, 2, 0).reshape(kh * kw, c).to(torch.float32).to(device).contiguous()     # [tap][C]
        self.bias = b.to(torch.float32).to(device).contiguous()

    def out_hw(self, h, w):
        return ((h + self.pad0 + self.pad1 - self.k) // self.stride + 1, (w + self.pad0 + self.pad1 - self.k) // self.stride + 1)


def depthwise_conv(x, plan):
    """x (N, H, W, C) channels-last in plan.dtype -> (N, Ho, Wo, C)"""
    _need_cuda(x)
    n, h, w, c = x.shape
    if c != plan.c or x.dtype != plan.dtype or not x.is_contiguous():
        raise CobevtHipError("depthwise_conv: bad input %s %s for C=%d %s" % (tuple(x.shape), x.dtype, plan.c, plan.dtype))
    ho, wo = plan.out_hw(h, w)
    out = torch.empty((n, ho, wo, c), device=x.device, dtype=plan.dtype)
    dims = _ints([plan.code, n, h, w, c, plan.k, plan.stride, plan.pad0, plan.pad0, ho, wo, plan.act])

    def cost():
        esz = 2 if plan.code == BF16 else 4
        return 2.0 * n * ho * wo * c * plan.k * plan.k, float((x.numel() + out.numel()) * esz)

    with _timed("depthwise|%dx%d s%d C%d %dx%d" % (plan.k, plan.k, plan.stride, c, h, w), cost):
        rc = _L.load().cobevt_depthwise_conv_nhwc(_p(x), _p(plan.wgt), _p(plan.bias), _p(out), dims, _stream())
    _L.check(rc, "cobevt_depthwise_conv_nhwc")
    return out


def spatial_mean(x):
    """(N, H, W, C) -> (N, C) fp32 mean over the pixels (the squeeze of squeeze-and-excitation), deterministic order"""
    _need_cuda(x)
    n, h, w, c = x.shape
    if not x.is_contiguous():
        raise CobevtHipError("spatial_mean: input must be contiguous channels-last")
    out = torch.empty((n, c), device=x.device, dtype=torch.float32)
    rc = _L.load().cobevt_spatial_mean_nhwc(_p(x), _p(out), dcode(x.dtype), n, h * w, c, _stream())
    _L.check(rc, "cobevt_spatial_mean_nhwc")
    return out


def se_gate(mean, w_reduce, b_reduce, w_expand, b_expand):
    """mean (N, C) fp32; w_reduce (Cs, C), b_reduce (Cs,), w_expand (C, Cs), b_expand (C,) fp32 ->
    sigmoid(w_expand . swish(w_reduce . mean + b_reduce) + b_expand)  (N, C) fp32"""
    _need_cuda(mean, w_reduce, b_reduce, w_expand, b_expand)
    n, c = mean.shape
    cs = w_reduce.shape[0]
    if tuple(w_reduce.shape) != (cs, c) or tuple(w_expand.shape) != (c, cs) or b_reduce.numel() != cs or b_expand.numel() != c:
        raise CobevtHipError("se_gate: inconsistent shapes")
    gate = torch.empty((n, c), device=mean.device, dtype=torch.float32)
    rc = _L.load().cobevt_se_gate(_p(mean), _p(w_reduce), _p(b_reduce), _p(w_expand), _p(b_expand), _p(gate), n, c, cs, _stream())
    _L.check(rc, "cobevt_se_gate")
    return gate


def channel_gate(x, gate):
    """x (N, H, W, C) * gate (N, C) fp32 -> (N, H, W, C)"""
    _need_cuda(x, gate)
    n, h, w, c = x.shape
    if tuple(gate.shape) != (n, c) or gate.dtype != torch.float32 or not x.is_contiguous() or not gate.is_contiguous():
        raise CobevtHipError("channel_gate: gate must be (N, C) fp32 for a contiguous (N, H, W, C) map")
    out = torch.empty_like(x)
    rc = _L.load().cobevt_channel_gate_nhwc(_p(x), _p(gate), _p(out), dcode(x.dtype), n, h * w, c, _stream())
    _L.check(rc, "cobevt_channel_gate_nhwc")
    return out


_DEFERRED_LABELS = []


_LABEL_RING = 32


def _label_word(c):
    """The next pinned host word of a ring of _LABEL_RING (entries of _DEFERRED_LABELS are reused round-robin; _check_deferred_labels has
    just read them all).  Pinned memory cannot be allocated while the stream is being captured into a HIP graph: the ring is filled by
    the eager calls before a capture (CapturedTrainStep's warm-up steps); a capture that finds it short skips the copy (None)."""
    global _LABEL_NEXT
    if len(_DEFERRED_LABELS) < _LABEL_RING:
        if torch.cuda.is_current_stream_capturing():
            if not _DEFERRED_LABELS:
                return None
        else:
            _DEFERRED_LABELS.append([torch.zeros(1, dtype=torch.float32).pin_memory(), c])
            return _DEFERRED_LABELS[-1][0]
    _LABEL_NEXT = (_LABEL_NEXT + 1) % len(_DEFERRED_LABELS)
    ent = _DEFERRED_LABELS[_LABEL_NEXT]
    ent[1] = c
    return ent[0]


_LABEL_NEXT = -1


def _check_deferred_labels():
    for word, c in _DEFERRED_LABELS:
        bad = int(word[0])
        if bad:
            word[0] = 0.0
            raise CobevtHipError("weighted_cross_entropy: %d target labels outside [0, %d) in an earlier call (only -100 is ignored)"
                                 % (bad, c))


def check_deferred_label_errors():
    """synchronise and raise if any weighted_cross_entropy call so far saw labels outside [0, C) (other than -100)"""
    torch.cuda.synchronize()
    _check_deferred_labels()


def weighted_cross_entropy(logits, target, weight, want_stats=False):
    """nn.CrossEntropyLoss(weight=weight)(logits (N, C, H, W), target (N, H, W) int64) -> 0-d fp32 tensor on the device
    (vanilla_seg_loss.py:18-23,58-70); forward only."""
    _need_cuda(logits, target)
    if logits.dim() != 4 or tuple(target.shape) != (logits.shape[0],) + tuple(logits.shape[2:]):
        raise CobevtHipError("weighted_cross_entropy expects (N, C, H, W) logits and an (N, H, W) target")
    x, y = logits.contiguous(), target.to(torch.int64).contiguous()
    n, c, h, w = x.shape
    wt = weight.to(device=x.device, dtype=torch.float32).contiguous()
    if wt.numel() != c:
        raise CobevtHipError("weighted_cross_entropy: %d class weights for %d classes" % (wt.numel(), c))
    scratch = torch.empty(3 * n * ((h * w + 4095) // 4096), device=x.device, dtype=torch.float32)
    out = torch.empty(4, device=x.device, dtype=torch.float32)
    rc = _L.load().cobevt_weighted_cross_entropy(_p(x), _p(y), _p(wt), _p(scratch), _p(out), dcode(x.dtype), n, c, h * w, _stream())
    _L.check(rc, "cobevt_weighted_cross_entropy")
    # nn.CrossEntropyLoss raises for targets outside [0, C) other than ignore_index = -100.  Here the count of such labels leaves the
    # device WITHOUT a host sync (a blocking read would drain the launch queue twice per training step): it is copied into a pinned
    # host word behind the kernel and inspected by the following calls (and by check_deferred_label_errors()), so a bad label
    # raises one call late - identically in eager steps and in steps replayed from a captured HIP graph (the copy is a graph node)
    _check_deferred_labels()
    host_word = _label_word(c)
    if host_word is not None:
        host_word.copy_(out[3:4], non_blocking=True)
    return (out[0], out, x, y, wt) if want_stats else out[0]


def iou_counts(pred, label, visibility, label_indices, thresholds, min_visibility):
    """(T, 3) int64 host tensor of (tp, fp, fn) per threshold; nuscenes metrics.py:22-31,56-72.  pred (N, C, hw) fp32 logits on
    the device, label (N, NL, hw), visibility (N, hw) uint8 / None, label_indices: list (length C) of lists of label channels."""
    _need_cuda(pred)
    n, c, hw = pred.shape
    dev = pred.device
    label = label.to(device=dev, dtype=torch.float32).contiguous()
    nl = label.shape[1]
    if len(label_indices) != c or nl > 32 or tuple(label.shape) != (n, nl, hw):
        raise CobevtHipError("iou_counts: %d prediction channels need %d label groups over at most 32 label channels" % (c, c))
    masks = torch.tensor([sum(1 << int(l) for l in group) for group in label_indices], dtype=torch.int64).to(torch.int32).to(dev)
    thr = torch.as_tensor(thresholds, dtype=torch.float32).to(dev).contiguous()
    counts = torch.zeros((thr.numel(), 3), device=dev, dtype=torch.int64)
    vis = None
    mv = -1
    if min_visibility is not None:
        vis = visibility.to(device=dev, dtype=torch.uint8).contiguous()
        mv = int(min_visibility)
    rc = _L.load().cobevt_iou_counts(_p(pred.contiguous()), _p(label), _p(vis), _p(masks), _p(thr), _p(counts), n, c, nl, hw,
                                     thr.numel(), mv, _stream())
    _L.check(rc, "cobevt_iou_counts")
    return counts.cpu()


def sigmoid_focal_loss_mean(pred, label, visibility, label_indices, min_visibility, alpha, gamma):
    """mean sigmoid focal loss over the visible pixels (nuscenes losses.py:27-84, forward).  pred (N, C, hw) fp32 logits on the
    device; label (N, NL, hw); label_indices: list (length C) of label-channel lists, or None = use label channel c as is."""
    _need_cuda(pred)
    n, c, hw = pred.shape
    dev = pred.device
    label = label.to(device=dev, dtype=torch.float32).contiguous()
    nl = label.shape[1]
    soft = label_indices is None
    if (soft and nl != c) or (not soft and len(label_indices) != c) or nl > 32 or tuple(label.shape) != (n, nl, hw):
        raise CobevtHipError("sigmoid_focal_loss_mean: inconsistent prediction / label channels")
    masks = None
    if not soft:
        masks = torch.tensor([sum(1 << int(l) for l in g) for g in label_indices], dtype=torch.int64).to(torch.int32).to(dev)
    vis, mv = None, -1
    if min_visibility is not None:
        vis, mv = visibility.to(device=dev, dtype=torch.uint8).contiguous(), int(min_visibility)
    scratch = torch.empty(2 * n * ((hw + 2047) // 2048), device=dev, dtype=torch.float32)
    out = torch.empty(3, device=dev, dtype=torch.float32)
    rc = _L.load().cobevt_sigmoid_focal_loss(_p(pred.float().contiguous()), _p(label), _p(vis), _p(masks), _p(scratch), _p(out), n, c, nl,
                                             hw, mv, ctypes.c_float(alpha), ctypes.c_float(gamma), int(soft), _stream())
    _L.check(rc, "cobevt_sigmoid_focal_loss")
    return out[0]
