"""Generate the golden vectors by running the REFERENCE (read-only, /root/reference) in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/gv*.npz

The reference is imported as Python with two stand-ins for absent third-party packages (tests/golden/_standins.py).
Only OUTPUT tensors (and tiny integer maps) are stored: inputs and weights are procedural
(cobevt_amd.synth), see tests/golden/cases.py.  Every case is also checked against the oracle on the spot so a
mismatch between reference and restatement is caught at generation time.  The fixtures travel to the GPU box;
the reference never does.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _standins  # noqa: E402

_standins.install()
sys.path.insert(0, "/root/reference/opv2v")

import cases  # noqa: E402
from cobevt_amd import synth  # noqa: E402
from cobevt_amd.synth import fill_module_  # noqa: E402
import oracle.corpbevt as o_model  # noqa: E402
import oracle.fax as o_fax  # noqa: E402
import oracle.resnet as o_resnet  # noqa: E402
import oracle.sttf as o_sttf  # noqa: E402
import oracle.swap_fusion as o_swap  # noqa: E402

from einops import rearrange  # noqa: E402
from opencood.models.sub_modules import fax_modules as R_fax  # noqa: E402
from opencood.models.fusion_modules import swap_fusion_modules as R_swap  # noqa: E402
from opencood.models.sub_modules import torch_transformation_utils as R_ttu  # noqa: E402
from opencood.models.sub_modules.fuse_utils import regroup as R_regroup  # noqa: E402
from opencood.models.sub_modules.naive_decoder import NaiveDecoder as R_NaiveDecoder  # noqa: E402
from opencood.models.sub_modules.bev_seg_head import BevSegHead as R_BevSegHead  # noqa: E402
from opencood.models.backbones.resnet_ms import ResnetEncoder as R_ResnetEncoder  # noqa: E402
from opencood.models import corpbevt as R_corpbevt  # noqa: E402
from opencood.models.fax_fused_transformer import FaxFusedTransformer as R_FaxFused  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def _np(t):
    return t.detach().cpu().numpy()


def _close(name, got, ref, tol=2e-5):
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    print("  oracle vs reference %-34s max|err| %.3e  (max|ref| %.3e)" % (name, err, scale))
    assert err <= tol * max(1.0, scale), "%s: oracle disagrees with the reference" % name


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def gv0():
    """state_dict key/shape schema of the reference models (drop-in contract)."""
    import copy
    out = {}
    for name, cfg, cls in (("corpbevt_full", synth.corpbevt_config(), R_corpbevt.CorpBEVT),
                           ("corpbevt_small", synth.corpbevt_small_config(), R_corpbevt.CorpBEVT)):
        m = cls(copy.deepcopy(cfg))
        sd = m.state_dict()
        out[name + "_keys"] = np.array(list(sd.keys()))
        out[name + "_shapes"] = np.array([",".join(str(int(d)) for d in v.shape) for v in sd.values()])
    save("gv0_state_dict_schema", **out)


def gv1():
    out = {}
    for (H, W, w1, w2) in cases.INDEX_MAP_SHAPES:
        a = torch.arange(H * W).reshape(H, W)
        win = rearrange(a, "(x w1) (y w2) -> (x y) (w1 w2)", w1=w1, w2=w2)
        grd = rearrange(a, "(w1 x) (w2 y) -> (x y) (w1 w2)", w1=w1, w2=w2)
        assert np.array_equal(_np(win), o_fax.window_partition_index(H, W, w1, w2))
        assert np.array_equal(_np(grd), o_fax.grid_partition_index(H, W, w1, w2))
        out["win_%d_%d_%d_%d" % (H, W, w1, w2)] = _np(win).astype(np.int32)
        out["grid_%d_%d_%d_%d" % (H, W, w1, w2)] = _np(grd).astype(np.int32)
    for (L, w) in cases.REL_POS_3D:
        idx = _np(R_swap.Attention(32, 32, 0.0, L, w).relative_position_index)
        assert np.array_equal(idx, o_swap.relative_position_index_3d(L, w))
        out["rel3d_%d_%d" % (L, w)] = idx.astype(np.int32)
    for w in cases.REL_POS_2D:
        idx = _np(R_fax.Attention(32, 32, 0.0, w).rel_pos_indices)
        assert np.array_equal(idx, o_fax.rel_pos_index_2d(w))
        if w <= 8:
            out["rel2d_%d" % w] = idx.astype(np.int32)
        else:  # 1024 x 1024: keep a strided sample and a checksum
            out["rel2d_%d_sample" % w] = idx[::37, ::41].astype(np.int32)
            out["rel2d_%d_sum" % w] = np.array([idx.sum()], dtype=np.int64)
    save("gv1_index_maps", **out)


def gv2():
    out = {}
    for name, c in cases.CROSS_WIN.items():
        m = fill_module_(R_fax.CrossWinAttention(c["dim"], c["heads"], c["dim_head"], c["qkv_bias"]).eval(), cases.SEED)
        q, k, v, skip = cases.cross_win_inputs(name)
        ref = m(q, k, v, skip)
        got = o_fax.cross_win_attention(m.state_dict(), "", q, k, v, skip, c["heads"], c["dim_head"])
        _close("CrossWinAttention." + name, got, ref)
        out[name] = _np(ref)
    save("gv2_cross_win_attention", **out)


def gv3():
    out = {}
    for name, c in cases.CVSA.items():
        fd, fh, fw = c["feat"]
        m = R_fax.CrossViewSwapAttention(fh, fw, fd, c["dim"], c["index"], c["image"][0], c["image"][1], **c["kwargs"]).eval()
        fill_module_(m, cases.SEED)
        bev = R_fax.BEVEmbedding(c["dim"], **c["bev_embedding"])
        x, feat, I_inv, E = cases.cvsa_inputs(name)
        ref = m(c["index"], x, bev, feat, I_inv, E)
        cfg = dict(c["kwargs"], image_height=c["image"][0], image_width=c["image"][1])
        grid = o_fax.bev_grids(**c["bev_embedding"])[c["index"]]
        assert torch.equal(grid, getattr(bev, "grid%d" % c["index"]))
        assert torch.equal(o_fax.image_plane(fh, fw, *c["image"]), m.image_plane[0, 0])
        got = o_fax.cross_view_swap_attention(m.state_dict(), "", cfg, c["index"], x, grid, feat, I_inv, E)
        _close("CrossViewSwapAttention." + name, got, ref)
        out[name] = _np(ref)
    save("gv3_cross_view_swap_attention", **out)


def gv4():
    c = cases.FAX_SMALL
    cfg = {k: (dict(v) if isinstance(v, dict) else list(v)) for k, v in c["config"].items()}
    m = fill_module_(R_fax.FAXModule(cfg).eval(), cases.SEED)
    batch = cases.fax_small_inputs()
    ref = m(batch)
    got = o_fax.fax_module(m.state_dict(), "", c["config"], batch["features"], batch["intrinsic"], batch["extrinsic"])
    _close("FAXModule.small", got, ref)
    save("gv4_fax_module", out=_np(ref))


def gv5():
    c = cases.SWAP
    x, mask = cases.swap_inputs()
    out = {}
    w = c["window_size"]
    att = fill_module_(R_swap.Attention(c["dim"], c["dim_head"], 0.1, c["agent_size"], w).eval(), cases.SEED)
    xw = rearrange(x, "b m d (x w1) (y w2) -> b m x y w1 w2 d", w1=w, w2=w)
    mw = rearrange(mask, "b (x w1) (y w2) e l -> b x y w1 w2 e l", w1=w, w2=w)
    ref = att(xw, mask=mw)
    got = o_swap.swap_attention(att.state_dict(), "", xw, mw, c["dim_head"], c["agent_size"], w)
    _close("swap Attention (window, mask)", got, ref)
    out["attention_window_mask"] = _np(ref)
    ref = att(xw, mask=None)
    _close("swap Attention (window, no mask)", o_swap.swap_attention(att.state_dict(), "", xw, None, c["dim_head"], c["agent_size"], w), ref)
    out["attention_window_nomask"] = _np(ref)

    blk = fill_module_(R_swap.SwapFusionBlockMask(c["dim"], c["mlp_dim"], c["dim_head"], w, c["agent_size"], 0.1).eval(), cases.SEED)
    ref = blk(x, mask)
    names = ["window_attention.", "window_ffd.", "grid_attention.", "grid_ffd."]
    _close("SwapFusionBlockMask", o_swap.swap_fusion_block(blk.state_dict(), names, x, mask, c["dim_head"], c["agent_size"], w), ref)
    out["block_mask"] = _np(ref)

    for use_mask in (True, False):
        args = dict(input_dim=c["dim"], mlp_dim=c["mlp_dim"], agent_size=c["agent_size"], window_size=w,
                    dim_head=c["dim_head"], drop_out=0.1, depth=c["depth"], mask=use_mask)
        enc = fill_module_(R_swap.SwapFusionEncoder(args).eval(), cases.SEED)
        ref = enc(x, mask if use_mask else None)
        got = o_swap.swap_fusion_encoder(enc.state_dict(), "", args, x, mask if use_mask else None)
        _close("SwapFusionEncoder mask=%s" % use_mask, got, ref)
        out["encoder_mask" if use_mask else "encoder_nomask"] = _np(ref)
    save("gv5_swap_fusion", **out)


def gv6():
    out = {}
    s = cases.STTF
    for (h, w) in ((16, 16), (12, 16)):
        x, tm, cav = cases.sttf_inputs(h, w)
        m = R_corpbevt.STTF({"resolution": s["resolution"], "downsample_rate": s["downsample_rate"]})
        ref = m(x, tm.clone())
        got = o_sttf.sttf(x, tm, s["resolution"], s["downsample_rate"])
        _close("STTF %dx%d" % (h, w), got, ref)
        out["sttf_%dx%d" % (h, w)] = _np(ref)
        refm = R_ttu.get_roi_and_cav_mask(tuple(ref.shape), cav, tm.clone(), s["resolution"], s["downsample_rate"])
        gotm = o_sttf.roi_and_cav_mask(tuple(ref.shape), cav, tm, s["resolution"], s["downsample_rate"])
        assert torch.equal(refm.float(), gotm.float()), "ROI mask mismatch"
        out["mask_%dx%d" % (h, w)] = _np(refm.float())
    dense = synth.procedural_input("gv6.regroup", (5, 4, 6, 6), cases.SEED)
    rl = torch.tensor([2, 3])
    rg, rmask = R_regroup(dense, rl, 3)
    og, omask = o_sttf.regroup(dense, rl, 3)
    assert torch.equal(rg, og) and torch.equal(rmask, omask)
    out["regroup"] = _np(rg)
    out["regroup_mask"] = _np(rmask).astype(np.int32)
    save("gv6_sttf_regroup", **out)


def gv7():
    out = {}
    d = cases.DECODER
    dec = fill_module_(R_NaiveDecoder(dict(d)).eval(), cases.SEED)
    x = synth.procedural_input("gv7.x", (1, 2, d["input_dim"], 8, 8), cases.SEED)
    ref = dec(x)
    _close("NaiveDecoder", o_model.naive_decoder(dec.state_dict(), "", d, x), ref)
    out["decoder"] = _np(ref)
    y = ref.reshape(-1, *ref.shape[2:])
    for target, classes in (("dynamic", 2), ("static", 3), ("both", 2)):
        head = fill_module_(R_BevSegHead(target, d["num_ch_dec"][0], classes).eval(), cases.SEED)
        r = head(y, 1, 2)
        g = o_model.bev_seg_head(head.state_dict(), "", target, y, 1, 2)
        for key in ("static_seg", "dynamic_seg"):
            _close("BevSegHead.%s.%s" % (target, key), g[key], r[key])
            out["head_%s_%s" % (target, key)] = _np(r[key])
    save("gv7_decoder_head", **out)


def gv8():
    cfg = synth.corpbevt_small_config()
    import copy
    m = R_corpbevt.CorpBEVT(copy.deepcopy(cfg)).eval()
    fill_module_(m, cases.SEED)
    batch = synth.opv2v_batch(agents=2, cams=2, image=128, max_cav=3, seed=cases.SEED)
    inter = {}
    m.fax.register_forward_hook(lambda mod, i, o: inter.__setitem__("fax", o))
    m.fusion_net.register_forward_hook(lambda mod, i, o: inter.__setitem__("fused", o))
    ref = m({k: v.clone() for k, v in batch.items()})
    got = o_model.corpbevt_forward(m.state_dict(), cfg, batch, return_intermediates=True)
    _close("CorpBEVT.small fax", got["fax"], inter["fax"].squeeze(1))
    _close("CorpBEVT.small fused", got["fused"], inter["fused"])
    _close("CorpBEVT.small dynamic_seg", got["dynamic_seg"], ref["dynamic_seg"])
    assert torch.equal(got["static_seg"], ref["static_seg"])
    nkeys = len(m.state_dict())
    save("gv8_corpbevt_small", dynamic_seg=_np(ref["dynamic_seg"]), static_seg=_np(ref["static_seg"]),
         fax=_np(inter["fax"]), fused=_np(inter["fused"]),
         argmax=_np(ref["dynamic_seg"].argmax(2)).astype(np.int8), n_state_dict_keys=np.array([nkeys]))

    # FaxFusedTransformer (SinBEVT on OPV2V, no fusion) on the same reduced config
    cfg2 = {k: copy.deepcopy(v) for k, v in cfg.items() if k in ("target", "encoder", "decoder", "fax", "seg_head_dim", "output_class")}
    m2 = fill_module_(R_FaxFused(copy.deepcopy(cfg2)).eval(), cases.SEED)
    b2 = {k: batch[k].reshape(1, 2, *batch[k].shape[2:]) for k in ("inputs", "intrinsic", "extrinsic")}
    ref2 = m2({k: v.clone() for k, v in b2.items()})
    got2 = o_model.fax_fused_transformer_forward(m2.state_dict(), cfg2, b2)
    _close("FaxFusedTransformer.small dynamic_seg", got2["dynamic_seg"], ref2["dynamic_seg"])
    save("gv8_fax_fused_small", dynamic_seg=_np(ref2["dynamic_seg"]))


def gv9():
    c = cases.GLOBAL_ATTN
    m = fill_module_(R_fax.Attention(c["dim"], c["dim_head"], 0.1, c["window_size"]).eval(), cases.SEED)
    x = synth.procedural_input("gv9.x", (c["b"], c["dim"], c["window_size"], c["window_size"]), cases.SEED)
    ref = m(x)
    _close("FAX global Attention", o_fax.global_attention(m.state_dict(), "", x, c["dim_head"], c["window_size"]), ref)
    save("gv9_global_attention", out=_np(ref))


def gv10():
    out = {}
    for depth, cfg in cases.RESNET.items():
        m = fill_module_(R_ResnetEncoder(dict(cfg)).eval(), cases.SEED)
        x = synth.procedural_input("gv10.x", (1, 1, 2, 64, 64, 3), cases.SEED)
        ref = m(x)
        got = o_resnet.resnet_encoder(m.state_dict(), "encoder.", cfg, x)
        for i, (r, g) in enumerate(zip(ref, got)):
            _close("ResnetEncoder%d[%d]" % (depth, i), g, r)
            out["resnet%d_f%d" % (depth, i)] = _np(r)
        out["resnet%d_shapes" % depth] = np.array([list(s) for s in m.output_shapes], dtype=np.int32)
    save("gv10_resnet_encoder", **out)


def gv11():
    """nuScenes SinBEVT: reference PyramidAxialEncoder + Decoder + CrossViewTransformer on the real config shapes."""
    sys.path.insert(0, "/root/reference/nuscenes")
    from cross_view_transformer.model.encoder_pyramid_axial import PyramidAxialEncoder as R_Enc
    from cross_view_transformer.model.decoder import Decoder as R_Dec
    from cross_view_transformer.model.cvt import CrossViewTransformer as R_CVT
    from cobevt_amd.synth import FeatureMapBackbone
    import copy
    import oracle.nuscenes as o_nu
    c = cases.NUSCENES
    feats, image, intr, ext = cases.nuscenes_inputs()
    enc = R_Enc(FeatureMapBackbone(feats), **copy.deepcopy(c["encoder"]))
    model = R_CVT(enc, R_Dec(**c["decoder"]), c["dim_last"], c["outputs"]).eval()
    fill_module_(model, cases.SEED)
    inter = {}
    model.encoder.register_forward_hook(lambda mod, i, o: inter.__setitem__("enc", o))
    model.decoder.register_forward_hook(lambda mod, i, o: inter.__setitem__("dec", o))
    ref = model({"image": image, "intrinsics": intr, "extrinsics": ext})
    sd = model.state_dict()
    got_enc = o_nu.pyramid_axial_encoder(sd, "encoder.", c["encoder"], feats, intr, ext)
    _close("nuScenes PyramidAxialEncoder", got_enc, inter["enc"])
    got = o_nu.cross_view_transformer(sd, c["encoder"], len(c["decoder"]["blocks"]), c["outputs"], feats, intr, ext)
    for k in ref:
        _close("nuScenes CrossViewTransformer[%s]" % k, got[k], ref[k])
    nrm = model.encoder.norm(image.flatten(0, 1))
    assert torch.allclose(nrm, o_nu.normalize(image.flatten(0, 1)))
    save("gv11_nuscenes_sinbevt", encoder=_np(inter["enc"]), bev=_np(ref["bev"]), center=_np(ref["center"]),
         normalized_image_sample=_np(nrm[:, :, ::37, ::41]),
         keys=np.array(list(sd.keys())), shapes=np.array([",".join(str(int(d)) for d in v.shape) for v in sd.values()]))


def gv12():
    """Pre-processor / collate / post-processor / scores, from the reference's own functions.
    Their modules import packages this image lacks (cv2, timm, open3d, matplotlib ...): those imports are satisfied with
    inert mock modules - none of the functions replayed here calls into them (cv2.resize / cvtColor paths are NOT pinned)."""
    import importlib
    import tempfile
    from unittest import mock
    import torch.nn as nn
    import oracle.pre_post as o_pp
    for name in ("cv2", "timm", "timm.scheduler", "timm.scheduler.cosine_lr", "open3d", "matplotlib", "matplotlib.pyplot",
                 "tensorboardX"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = mock.MagicMock(name=name)
    from opencood.data_utils.pre_processor.rgb_preprocessor import RgbPreProcessor as R_Rgb
    from opencood.data_utils.post_processor.camera_bev_postprocessor import CameraBevPostprocessor as R_Post
    from opencood.utils import seg_utils as R_seg
    from opencood.tools import train_utils as R_train
    from opencood.data_utils.datasets.camera_only.intermediate_fusion_dataset import CamIntermediateFusionDataset as R_Data

    c, inp, out = cases.PRE_POST, cases.pre_post_inputs(), {}
    # pre-processor (no channel swap / resize: both are cv2 calls)
    pre = R_Rgb({"args": {"mean": c["mean"], "std": c["std"], "bgr2rgb": False, "resize_x": 0, "resize_y": 0}}, train=False)
    std_img = pre.standalize(pre.normalize(inp["image_u8"]))
    assert np.array_equal(std_img, o_pp.standardize_rgb(inp["image_u8"], c["mean"], c["std"], False))
    out["standardized"] = std_img
    # post-processor
    post = R_Post({}, train=False)
    logits = {k: torch.from_numpy(inp[k + "_logits"]) for k in ("static", "dynamic")}
    ref = post.post_process_train({"static_seg": logits["static"][:, None], "dynamic_seg": logits["dynamic"][:, None]})
    mine = o_pp.post_process_train({"static_seg": logits["static"][:, None], "dynamic_seg": logits["dynamic"][:, None]})
    for k in ("static_prob", "static_map", "dynamic_prob", "dynamic_map"):
        assert torch.equal(ref[k], mine[k]), k
        out[k] = _np(ref[k])
    merged = post.merge_label(inp["road"], inp["lane"])
    assert np.array_equal(merged, o_pp.merge_label(inp["road"], inp["lane"]))
    out["merged"] = merged
    # scores
    for i, (pred, gt) in enumerate(inp["pairs"]):
        iu, mp = R_seg.mean_IU(pred, gt), R_seg.mean_precision(pred, gt)
        assert list(iu) == o_pp.mean_iu(pred, gt) and list(mp) == o_pp.mean_precision(pred, gt)
        out["iu%d" % i], out["precision%d" % i] = np.array(iu, dtype=np.float64), np.array(mp, dtype=np.float64)
    # collate
    holder = types.SimpleNamespace(train=True)
    ref_b = R_Data.collate_batch(holder, inp["samples"])["ego"]
    my_b = o_pp.collate_batch(inp["samples"], train=True)["ego"]
    for k, v in ref_b.items():
        assert v.dtype == my_b[k].dtype and torch.equal(v, my_b[k]), k
        out["collate_" + k] = _np(v)
        out["collate_dtype_" + k] = np.array(str(v.dtype))
    # checkpoint discovery + non-strict load
    with tempfile.TemporaryDirectory() as d:
        net = nn.Linear(3, 2)
        for ep in (3, 12, 7):
            torch.save({"weight": torch.full((2, 3), float(ep)), "stray.key": torch.zeros(1)}, os.path.join(d, "net_epoch%d.pth" % ep))
        ep, net = R_train.load_saved_model(d, net)
        ep2, net2 = o_pp.load_saved_model(d, nn.Linear(3, 2))
        assert ep == ep2 == 12 and torch.equal(net.weight, net2.weight)
        out["loaded_epoch"], out["loaded_weight"] = np.array(ep), _np(net.weight)
    with tempfile.TemporaryDirectory() as d:
        ep, _ = R_train.load_saved_model(d, nn.Linear(3, 2))
        assert ep == 0 == o_pp.load_saved_model(d, nn.Linear(3, 2))[0]
        out["empty_epoch"] = np.array(ep)
    save("gv12_pre_post", **out)


def gv14():
    """VanillaSegLoss (loss/vanilla_seg_loss.py) forward on the three target modes.  Its constructor moves the class weights
    with .cuda(); there is no GPU in the build container, so Tensor.cuda is an identity for the duration of the call."""
    import importlib
    from unittest import mock
    import oracle.pre_post as o_pp
    for name in ("cv2", "timm", "timm.scheduler", "timm.scheduler.cosine_lr", "open3d", "matplotlib", "matplotlib.pyplot", "tensorboardX"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = mock.MagicMock(name=name)
    from opencood.loss.vanilla_seg_loss import VanillaSegLoss as R_Loss
    inp = {k: torch.from_numpy(v) for k, v in cases.seg_loss_inputs().items()}
    out = {}
    for i, args in enumerate(cases.SEG_LOSS):
        with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
            crit = R_Loss(dict(args))
        total = crit({"static_seg": inp["static_seg"], "dynamic_seg": inp["dynamic_seg"]}, inp)
        mine = o_pp.vanilla_seg_loss(args, inp, inp)
        for k in ("total_loss", "static_loss", "dynamic_loss"):
            assert abs(float(crit.loss_dict[k]) - float(mine[k])) <= 1e-6 * max(1.0, abs(float(mine[k]))), (k, args)
            out["%s%d" % (k, i)] = np.array(float(crit.loss_dict[k]), dtype=np.float64)
        assert float(total) == float(crit.loss_dict["total_loss"])
        print("  VanillaSegLoss %-8s total %.6f static %.6f dynamic %.6f" % (args["target"], float(total), float(crit.loss_dict["static_loss"]),
                                                                             float(crit.loss_dict["dynamic_loss"])))
    save("gv14_vanilla_seg_loss", **out)


def gv15():
    """nuScenes IoUMetric (cross_view_transformer/metrics.py) over two update() calls, with the vehicle experiment's label
    grouping + visibility mask and with a two-channel grouping."""
    import importlib.util
    import oracle.pre_post as o_pp
    _standins.install_torchmetrics()
    spec = importlib.util.spec_from_file_location("ref_nuscenes_metrics", "/root/reference/nuscenes/cross_view_transformer/metrics.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    out = {}
    for i, c in enumerate(cases.IOU_METRIC):
        ups = [(torch.from_numpy(p), {"bev": torch.from_numpy(b), "visibility": torch.from_numpy(v)}) for p, b, v in cases.iou_metric_inputs(c["channels"])]
        m = R.IoUMetric(c["label_indices"], c["min_visibility"])
        for pred, batch in ups:
            m.update({"bev": pred}, batch)
        res = m.compute()
        tp, fp, fn, mine = o_pp.iou_metric(ups, c["label_indices"], c["min_visibility"])
        assert torch.equal(tp, m.tp) and torch.equal(fp, m.fp) and torch.equal(fn, m.fn) and mine == res, (mine, res)
        out["tp%d" % i], out["fp%d" % i], out["fn%d" % i] = _np(m.tp), _np(m.fp), _np(m.fn)
        out["iou%d" % i] = np.array([res[k] for k in sorted(res)], dtype=np.float64)
        print("  IoUMetric %s -> %s" % (c, res))
    save("gv15_nuscenes_iou_metric", **out)


def gv16():
    """nuScenes BinarySegmentationLoss / CenterLoss forward (cross_view_transformer/losses.py) with fvcore's sigmoid_focal_loss
    supplied by a restated stand-in."""
    import importlib.util
    import oracle.pre_post as o_pp
    _standins.install_fvcore()
    spec = importlib.util.spec_from_file_location("ref_nuscenes_losses", "/root/reference/nuscenes/cross_view_transformer/losses.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    inp = {k: torch.from_numpy(v) for k, v in cases.focal_loss_inputs().items()}
    batch = {"bev": inp["bev"], "center": inp["center"], "visibility": inp["visibility"]}
    out = {}
    for i, c in enumerate(cases.FOCAL_LOSS):
        if c["kind"] == "bev":
            ref = R.BinarySegmentationLoss(c["label_indices"], c["min_visibility"], c["alpha"], c["gamma"])({"bev": inp["bev_pred"]}, batch)
            mine = o_pp.binary_segmentation_loss({"bev": inp["bev_pred"]}, batch, c["label_indices"], c["min_visibility"], c["alpha"], c["gamma"])
        else:
            ref = R.CenterLoss(c["min_visibility"], c["alpha"], c["gamma"])({"center": inp["center_pred"]}, batch)
            mine = o_pp.center_loss({"center": inp["center_pred"]}, batch, c["min_visibility"], c["alpha"], c["gamma"])
        assert abs(float(ref) - float(mine)) <= 1e-7 * max(1.0, abs(float(ref))), (c, float(ref), float(mine))
        out["loss%d" % i] = np.array(float(ref), dtype=np.float64)
        print("  %s -> %.7f" % (c, float(ref)))
    total, parts = R.MultipleLoss({"bev": R.BinarySegmentationLoss([[4, 5]], 2), "bev_weight": 1.0,
                                   "center": R.CenterLoss(2), "center_weight": 0.1})({"bev": inp["bev_pred"], "center": inp["center_pred"]}, batch)
    out["multi_total"] = np.array(float(total), dtype=np.float64)
    save("gv16_nuscenes_losses", **out)


def gv13():
    """NaiveCompressor (sub_modules/naive_compress.py) alone and inside the reduced CorpBEVT with compression = 2."""
    import copy
    from opencood.models.sub_modules.naive_compress import NaiveCompressor as R_Comp
    comp = fill_module_(R_Comp(32, 4).eval(), cases.SEED)
    x = synth.procedural_input("gv13.x", (3, 32, 12, 16), cases.SEED, -2.0, 2.0)
    ref = comp(x)
    got = o_model.naive_compressor({"c." + k: v for k, v in comp.state_dict().items()}, "c.", x)
    _close("NaiveCompressor", got, ref)
    cfg = synth.corpbevt_small_compressed_config(2)
    m = R_corpbevt.CorpBEVT(copy.deepcopy(cfg)).eval()
    fill_module_(m, cases.SEED)
    batch = synth.opv2v_batch(agents=2, cams=2, image=128, max_cav=3, seed=cases.SEED)
    out = m({k: v.clone() for k, v in batch.items()})
    mine = o_model.corpbevt_forward(m.state_dict(), cfg, batch)
    _close("CorpBEVT.small compression=2", mine["dynamic_seg"], out["dynamic_seg"])
    save("gv13_naive_compressor", compressor=_np(ref), dynamic_seg=_np(out["dynamic_seg"]),
         keys=np.array([k for k in m.state_dict().keys() if k.startswith("naive_compressor.")]))


def gv17():
    """CVT baseline models at reduced size from the reference's own classes: CrossViewTransformer
    (single agent), CrossViewTransformerSwapFuse, CrossViewTransformerFcooper + the per-agent CrossViewModule output."""
    import copy
    import oracle.cvt as o_cvt
    from opencood.models.cross_view_transformer import CrossViewTransformer as R_Cvt
    from opencood.models.cross_view_transformer_swap_fuse import CrossViewTransformerSwapFuse as R_CvtSwap
    from opencood.models.cross_view_transformer_fcooper import CrossViewTransformerFcooper as R_CvtFcooper
    from opencood.models.cross_view_transformer_att_fuse import CrossViewTransformerAttFuse as R_CvtAtt
    from opencood.models.cross_view_transformer_v2vnet import CrossViewTransformerV2VNet as R_CvtV2V
    from opencood.models.cross_view_transformer_disconet import CrossViewTransformerDiscoNet as R_CvtDisco
    import oracle.v2v as o_v2v
    out = {}
    single = synth.opv2v_batch(agents=1, cams=2, image=128, max_cav=3, seed=cases.SEED)
    single_b = {k: single[k] for k in ("inputs", "intrinsic", "extrinsic")}
    multi = synth.opv2v_batch(agents=2, cams=2, image=128, max_cav=3, seed=cases.SEED)
    for kind, cls, fwd, batch in (("single", R_Cvt, o_cvt.cross_view_transformer_forward, single_b),
                                  ("swap_fuse", R_CvtSwap, o_cvt.cross_view_transformer_swap_fuse_forward, multi),
                                  ("fcooper", R_CvtFcooper, o_cvt.cross_view_transformer_fcooper_forward, multi),
                                  ("att_fuse", R_CvtAtt, o_cvt.cross_view_transformer_att_fuse_forward, multi),
                                  ("v2vnet", R_CvtV2V, o_v2v.cross_view_transformer_v2vnet_forward, multi),
                                  ("disconet", R_CvtDisco, o_v2v.cross_view_transformer_disconet_forward, multi)):
        cfg = synth.cvt_small_config(kind)
        m = fill_module_(cls(copy.deepcopy(cfg)).eval(), cases.SEED)
        ref = m(dict(batch))
        got = fwd(m.state_dict(), cfg, dict(batch))
        _close("CVT %s dynamic_seg" % kind, got["dynamic_seg"], ref["dynamic_seg"])
        out[kind + "_dynamic_seg"] = _np(ref["dynamic_seg"])
        out[kind + "_keys"] = np.array(list(m.state_dict().keys()))
        out[kind + "_shapes"] = np.array([",".join(str(int(d)) for d in v.shape) for v in m.state_dict().values()])
        if kind == "single":
            feats = m.encoder(batch["inputs"])
            cvm = m.cvm({"inputs": batch["inputs"], "intrinsic": batch["intrinsic"], "extrinsic": batch["extrinsic"], "features": feats})
            _close("CrossViewModule", o_cvt.encode_agents(m.state_dict(), cfg, dict(batch)), cvm)
            out["single_cvm"] = _np(cvm)
    save("gv17_cvt_baselines", **out)

GV18_SEEDS = (0, 1, 2, 3)
_AMP_DTYPE = torch.bfloat16          # gv19() re-runs gv18's cases under float16: the reference's OWN mixed-precision dtype


def _dev_stats(run, class_dim):
    """run(): reference forward returning a tensor, a list or {key: tensor}.  Evaluated in fp32 and again inside
    torch.autocast("cpu", dtype=torch.bfloat16) - the reference's own mixed-precision mode (train_camera.py:157-160 and
    nuscenes/scripts/benchmark.py:45 wrap the same forward in torch.cuda.amp.autocast).
    -> {output key: (max|d| / max|ref|, ||d||_2 / ||ref||_2, arg-max agreement along class_dim or 1.0)}"""
    ref = run()
    with torch.autocast("cpu", dtype=_AMP_DTYPE):
        amp = run()
    if torch.is_tensor(ref):
        ref, amp = {"": ref}, {"": amp}
    if isinstance(ref, (list, tuple)):
        ref, amp = {"[%d]" % i: v for i, v in enumerate(ref)}, {"[%d]" % i: v for i, v in enumerate(amp)}
    res = {}
    for k, r in ref.items():
        a, r = amp[k].float(), r.float()
        d = (a - r).double()
        mx = float(d.abs().max() / r.abs().max().clamp_min(1e-12))
        rms = float(d.square().sum().sqrt() / r.double().square().sum().sqrt().clamp_min(1e-30))
        agree = float((a.argmax(class_dim) == r.argmax(class_dim)).float().mean()) if class_dim is not None else 1.0
        res[k] = (mx, rms, agree)
    return res


def _dev(out, name, build, class_dim=None, seeds=GV18_SEEDS):
    """build(seed) -> run(): the reference module / model with the procedural weight set `seed` (fill_module_), on the inputs
    the tests use.  The bf16 deviation of a deep network is one realisation of accumulated rounding noise and moves by
    +-40 % between weight sets, so what is stored is the ENVELOPE over the weight sets 0..3 -
    [largest max-rel, largest rms-rel, smallest arg-max agreement] - plus the per-seed values under '<key>#seeds'."""
    per = [_dev_stats(build(sd), class_dim) for sd in seeds]
    for k in per[0]:
        key = name + (("." + k) if k and not k.startswith("[") else k)
        v = np.array([p[k] for p in per], dtype=np.float64)            # (seeds, 3)
        out[key] = np.array([v[:, 0].max(), v[:, 1].max(), v[:, 2].min()])
        out[key + "#seeds"] = v
        print("  reference " + str(_AMP_DTYPE).split(".")[-1] + "-autocast vs fp32 %-44s max-rel %.3e (%s)  rms-rel %.3e (%s)  arg-max agreement >= %.4f"
              % (key, out[key][0], " ".join("%.2e" % x for x in v[:, 0]), out[key][1], " ".join("%.2e" % x for x in v[:, 1]), out[key][2]))


def gv18(full=None, fixture="gv18_reference_bf16_autocast"):
    """The REFERENCE's own bf16 mixed-precision deviation: every module / model of GV2-GV11, GV13, GV17 and
    the full-size corpbevt.yaml frame run by the reference in fp32 and under torch.autocast(bfloat16) on the tests' procedural
    inputs, for the procedural weight sets 0..3.  These numbers - not anything measured on the HIP path - are what the bf16
    gates in tests/util.py are derived from: a bf16 result may deviate from the fp32 reference by as much as the reference's own
    bf16 run does, or 1e-2, whichever is larger."""
    import copy
    full = ("--no-full" not in sys.argv) if full is None else full
    out = {}
    for name, c in cases.CROSS_WIN.items():
        q, k, v, skip = cases.cross_win_inputs(name)

        def build(sd, c=c):
            m = fill_module_(R_fax.CrossWinAttention(c["dim"], c["heads"], c["dim_head"], c["qkv_bias"]).eval(), sd)
            return lambda: m(q, k, v, skip)
        _dev(out, "CrossWinAttention." + name, build)
    for name, c in cases.CVSA.items():
        fd, fh, fw = c["feat"]
        bev = R_fax.BEVEmbedding(c["dim"], **c["bev_embedding"])
        x, feat, I_inv, E = cases.cvsa_inputs(name)

        def build(sd, c=c):
            m = R_fax.CrossViewSwapAttention(fh, fw, fd, c["dim"], c["index"], c["image"][0], c["image"][1], **c["kwargs"]).eval()
            fill_module_(m, sd)
            return lambda: m(c["index"], x, bev, feat, I_inv, E)
        _dev(out, "CrossViewSwapAttention." + name, build)
    c = cases.FAX_SMALL
    batch = cases.fax_small_inputs()

    def build(sd):
        cfg = {k: (dict(v) if isinstance(v, dict) else list(v)) for k, v in c["config"].items()}
        m = fill_module_(R_fax.FAXModule(cfg).eval(), sd)
        return lambda: m(dict(batch))
    _dev(out, "FAXModule", build)
    c = cases.SWAP
    x, mask = cases.swap_inputs()
    w = c["window_size"]
    xw = rearrange(x, "b m d (x w1) (y w2) -> b m x y w1 w2 d", w1=w, w2=w)
    mw = rearrange(mask, "b (x w1) (y w2) e l -> b x y w1 w2 e l", w1=w, w2=w)
    for nm, mk in (("swap Attention + mask", mw), ("swap Attention", None)):
        def build(sd, mk=mk):
            att = fill_module_(R_swap.Attention(c["dim"], c["dim_head"], 0.1, c["agent_size"], w).eval(), sd)
            return lambda: att(xw, mask=mk)
        _dev(out, nm, build)

    def build(sd):
        blk = fill_module_(R_swap.SwapFusionBlockMask(c["dim"], c["mlp_dim"], c["dim_head"], w, c["agent_size"], 0.1).eval(), sd)
        return lambda: blk(x, mask)
    _dev(out, "SwapFusionBlockMask", build)
    for use_mask in (True, False):
        def build(sd, use_mask=use_mask):
            args = dict(input_dim=c["dim"], mlp_dim=c["mlp_dim"], agent_size=c["agent_size"], window_size=w,
                        dim_head=c["dim_head"], drop_out=0.1, depth=c["depth"], mask=use_mask)
            enc = fill_module_(R_swap.SwapFusionEncoder(args).eval(), sd)
            return lambda: enc(x, mask if use_mask else None)
        _dev(out, "SwapFusionEncoder mask=%s" % use_mask, build)
    # LiDAR-shaped FuseBEVT operator config (64 ch, 8 agents, window 8) on a 32 x 32 map
    xl = synth.procedural_input("gv18.lidar.x", (1, 8, 64, 32, 32), cases.SEED)
    ml = torch.ones(1, 32, 32, 1, 8)
    ml[0, :, :, :, 5:] = 0

    def build(sd):
        args = dict(input_dim=64, mlp_dim=128, agent_size=8, window_size=8, dim_head=32, drop_out=0.1, depth=3, mask=True)
        enc = fill_module_(R_swap.SwapFusionEncoder(args).eval(), sd)
        return lambda: enc(xl, ml)
    _dev(out, "SwapFusionEncoder lidar-shaped", build)
    d = cases.DECODER
    xd = synth.procedural_input("gv7.x", (1, 2, d["input_dim"], 8, 8), cases.SEED)

    def build(sd):
        dec = fill_module_(R_NaiveDecoder(dict(d)).eval(), sd)
        return lambda: dec(xd)
    _dev(out, "NaiveDecoder", build)
    dec0 = fill_module_(R_NaiveDecoder(dict(d)).eval(), cases.SEED)
    y = dec0(xd)
    y = y.reshape(-1, *y.shape[2:])
    for target, classes in (("dynamic", 2), ("static", 3), ("both", 2)):
        def build(sd, target=target, classes=classes):
            head = fill_module_(R_BevSegHead(target, d["num_ch_dec"][0], classes).eval(), sd)
            return lambda: {k: v for k, v in head(y, 1, 2).items() if v.is_floating_point() and v.numel() > 1 and float(v.abs().max()) > 0}
        _dev(out, "BevSegHead." + target, build, class_dim=2)
    c = cases.GLOBAL_ATTN
    xg = synth.procedural_input("gv9.x", (c["b"], c["dim"], c["window_size"], c["window_size"]), cases.SEED)

    def build(sd):
        m = fill_module_(R_fax.Attention(c["dim"], c["dim_head"], 0.1, c["window_size"]).eval(), sd)
        return lambda: m(xg)
    _dev(out, "FAX global attention", build)
    xr = synth.procedural_input("gv10.x", (1, 1, 2, 64, 64, 3), cases.SEED)
    for depth, cfg in cases.RESNET.items():
        def build(sd, cfg=cfg):
            m = fill_module_(R_ResnetEncoder(dict(cfg)).eval(), sd)
            return lambda: list(m(xr))
        _dev(out, "resnet%d" % depth, build)
    from opencood.models.sub_modules.naive_compress import NaiveCompressor as R_Comp
    xc = synth.procedural_input("gv13.x", (3, 32, 12, 16), cases.SEED, -2.0, 2.0)

    def build(sd):
        comp = fill_module_(R_Comp(32, 4).eval(), sd)
        return lambda: comp(xc)
    _dev(out, "NaiveCompressor", build)

    # reduced end-to-end models (GV8, GV13, GV17) on the tests' batches
    cfg = synth.corpbevt_small_config()
    batch = synth.opv2v_batch(agents=2, cams=2, image=128, max_cav=3, seed=cases.SEED)

    def build(sd):
        m = fill_module_(R_corpbevt.CorpBEVT(copy.deepcopy(cfg)).eval(), sd)
        inter = {}
        m.fax.register_forward_hook(lambda mod, i, o: inter.__setitem__("fax", o))
        m.fusion_net.register_forward_hook(lambda mod, i, o: inter.__setitem__("fused", o))

        def run():
            r = m({k: v.clone() for k, v in batch.items()})
            return {"dynamic_seg": r["dynamic_seg"], "fax": inter["fax"], "fused": inter["fused"]}
        return run
    _dev(out, "CorpBEVT.small", build, class_dim=2)
    # the ragged training-style batch of tests/test_modules_gpu.py::test_corpbevt_ragged_scenarios_vs_oracle
    record_len = [2, 1, 3]
    fullb = synth.opv2v_batch(agents=3, cams=2, image=128, max_cav=3, seed=cases.SEED + 1, batch=len(record_len))
    keep = [s_ * 3 + a for s_, n in enumerate(record_len) for a in range(n)]
    rb = {k: fullb[k][keep] for k in ("inputs", "intrinsic", "extrinsic")}
    rb["transformation_matrix"] = fullb["transformation_matrix"]
    rb["record_len"] = torch.tensor(record_len, dtype=torch.int64)

    def build(sd):
        m = fill_module_(R_corpbevt.CorpBEVT(copy.deepcopy(cfg)).eval(), sd)
        return lambda: m({k: v.clone() for k, v in rb.items()})["dynamic_seg"]
    _dev(out, "CorpBEVT ragged scenarios", build, class_dim=2)
    cfg2 = {k: copy.deepcopy(v) for k, v in cfg.items() if k in ("target", "encoder", "decoder", "fax", "seg_head_dim", "output_class")}
    b2 = {k: batch[k].reshape(1, 2, *batch[k].shape[2:]) for k in ("inputs", "intrinsic", "extrinsic")}

    def build(sd):
        m2 = fill_module_(R_FaxFused(copy.deepcopy(cfg2)).eval(), sd)
        return lambda: m2({k: v.clone() for k, v in b2.items()})["dynamic_seg"]
    _dev(out, "FaxFusedTransformer.small", build, class_dim=2)
    cfgc = synth.corpbevt_small_compressed_config(2)

    def build(sd):
        mc = fill_module_(R_corpbevt.CorpBEVT(copy.deepcopy(cfgc)).eval(), sd)
        return lambda: mc({k: v.clone() for k, v in batch.items()})["dynamic_seg"]
    _dev(out, "CorpBEVT.small compression=2", build, class_dim=2)

    from opencood.models.cross_view_transformer import CrossViewTransformer as R_Cvt
    from opencood.models.cross_view_transformer_swap_fuse import CrossViewTransformerSwapFuse as R_CvtSwap
    from opencood.models.cross_view_transformer_fcooper import CrossViewTransformerFcooper as R_CvtFcooper
    from opencood.models.cross_view_transformer_att_fuse import CrossViewTransformerAttFuse as R_CvtAtt
    from opencood.models.cross_view_transformer_v2vnet import CrossViewTransformerV2VNet as R_CvtV2V
    from opencood.models.cross_view_transformer_disconet import CrossViewTransformerDiscoNet as R_CvtDisco
    single = synth.opv2v_batch(agents=1, cams=2, image=128, max_cav=3, seed=cases.SEED)
    single_b = {k: single[k] for k in ("inputs", "intrinsic", "extrinsic")}
    for kind, cls, bt in (("single", R_Cvt, single_b), ("swap_fuse", R_CvtSwap, batch), ("fcooper", R_CvtFcooper, batch),
                          ("att_fuse", R_CvtAtt, batch), ("v2vnet", R_CvtV2V, batch), ("disconet", R_CvtDisco, batch)):
        def build(sd, kind=kind, cls=cls, bt=bt):
            mk = fill_module_(cls(copy.deepcopy(synth.cvt_small_config(kind))).eval(), sd)
            return lambda: mk(dict(bt))["dynamic_seg"]
        _dev(out, "CVT " + kind, build, class_dim=2)
        if kind == "single":
            def build(sd, cls=cls, bt=bt):
                mk = fill_module_(cls(copy.deepcopy(synth.cvt_small_config("single"))).eval(), sd)
                return lambda: mk.cvm({"inputs": bt["inputs"], "intrinsic": bt["intrinsic"], "extrinsic": bt["extrinsic"],
                                       "features": mk.encoder(bt["inputs"])})
            _dev(out, "CrossViewModule", build)

    # nuScenes SinBEVT at the real config shapes (GV11)
    sys.path.insert(0, "/root/reference/nuscenes")
    from cross_view_transformer.model.encoder_pyramid_axial import PyramidAxialEncoder as R_Enc
    from cross_view_transformer.model.decoder import Decoder as R_Dec
    from cross_view_transformer.model.cvt import CrossViewTransformer as R_CVT
    from cobevt_amd.synth import FeatureMapBackbone
    c = cases.NUSCENES
    feats, image, intr, ext = cases.nuscenes_inputs()

    def build(sd):
        enc = R_Enc(FeatureMapBackbone(feats), **copy.deepcopy(c["encoder"]))
        model = fill_module_(R_CVT(enc, R_Dec(**c["decoder"]), c["dim_last"], c["outputs"]).eval(), sd)
        inter_n = {}
        model.encoder.register_forward_hook(lambda mod, i, o: inter_n.__setitem__("enc", o))

        def run():
            r = dict(model({"image": image, "intrinsics": intr, "extrinsics": ext}))
            r["encoder"] = inter_n["enc"]
            return r
        return run
    _dev(out, "nuScenes SinBEVT", build)

    if full:
        # the full-size corpbevt.yaml frame (BASELINE configs[2] / [3] and the bench headline), inputs as in bench.py / the tests
        cfgf = synth.corpbevt_config()
        for agents in (2, 5):
            bf = synth.opv2v_batch(agents=agents, cams=4, image=512, max_cav=5, seed=0)

            def build(sd, bf=bf):
                mf = fill_module_(R_corpbevt.CorpBEVT(copy.deepcopy(cfgf)).eval(), sd)
                interf = {}
                mf.fax.register_forward_hook(lambda mod, i, o: interf.__setitem__("fax", o))
                mf.fusion_net.register_forward_hook(lambda mod, i, o: interf.__setitem__("fused", o))
                mf.encoder.register_forward_hook(lambda mod, i, o: interf.__setitem__("enc", o))
                mf.sttf.register_forward_hook(lambda mod, i, o: interf.__setitem__("sttf", o))
                for lv, layer in enumerate(mf.fax.layers):          # the level's BEV query in front of its down-sampling layer
                    layer.register_forward_hook(lambda mod, i, o, lv=lv: interf.__setitem__("fax_level%d" % lv, o))

                def run():
                    r = mf({k: v.clone() for k, v in bf.items()})
                    o = {"dynamic_seg": r["dynamic_seg"], "fax": interf["fax"], "fused": interf["fused"], "sttf": interf["sttf"]}
                    o.update({k: v for k, v in interf.items() if k.startswith("fax_level")})
                    for i, e in enumerate(interf["enc"]):
                        o["resnet34_f%d" % i] = e
                    return o
                return run
            _dev(out, "CorpBEVT.full %d agents" % agents, build, class_dim=2, seeds=GV18_SEEDS[:2])

            def build(sd, bf=bf, agents=agents):             # the bench frame: the class-balanced head of seed 0 (synth.balance_seg_head_)
                mf = fill_module_(R_corpbevt.CorpBEVT(copy.deepcopy(cfgf)).eval(), sd)
                synth.balance_seg_head_(mf, agents)
                return lambda: mf({k: v.clone() for k, v in bf.items()})["dynamic_seg"]
            _dev(out, "CorpBEVT.full %d agents balanced head" % agents, build, class_dim=2, seeds=(0,))
    save(fixture, **out)


def gv19():
    """the same cases as gv18 under torch.autocast("cpu", dtype=torch.float16) - float16 is the dtype the reference's
    own mixed-precision runs use (torch.cuda.amp.autocast in opv2v/opencood/tools/train_camera.py:157-160 and
    nuscenes/scripts/benchmark.py:45 defaults to float16 on a GPU).  A second yardstick beside gv18's bfloat16 one: how far the
    reference's fp16 run moves away from its fp32 forward.  Nothing in the test gates reads this fixture (there is no fp16 compute mode
    in the product); tests/test_oracle_golden.py checks that it is there and ordered as expected against gv18."""
    global _AMP_DTYPE
    prev = _AMP_DTYPE
    _AMP_DTYPE = torch.float16
    try:
        gv18(fixture="gv19_reference_fp16_autocast")
    finally:
        _AMP_DTYPE = prev


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["gv0", "gv1", "gv2", "gv3", "gv4", "gv5", "gv6", "gv7", "gv8", "gv9", "gv10", "gv11", "gv12", "gv13", "gv14", "gv15", "gv16", "gv17", "gv18"]
    for name in which:
        print("== " + name)
        globals()[name]()
