"""The data formats either side of the hot path: RgbPreProcessor, collate_batch,
CameraBevPostprocessor, seg_utils scores, load_saved_model.

CPU part: oracle/pre_post.py and the host mirrors replayed against tests/golden/gv12_pre_post.npz (outputs of the
REFERENCE's own functions, tests/golden/make_golden.py gv12).  Integer / index work bit-exact; float64 host arithmetic
bit-exact (same numpy formulas); softmax probabilities 1e-6 absolute (fp32 exp of a different libm).
GPU part (-m gpu): the HIP softmax+argmax and per-class count kernels through the C-ABI against the same fixtures."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

import cases
from cobevt_amd.host import camera_bev_postprocessor as h_post
from cobevt_amd.host import intermediate_fusion_dataset as h_data
from cobevt_amd.host import rgb_preprocessor as h_rgb
from cobevt_amd.host import seg_utils as h_seg
from cobevt_amd.host import train_utils as h_train
from cobevt_amd.lib import CobevtHipError
import oracle.pre_post as o_pp
from util import golden

PROB_TOL = 1e-6
torch.set_grad_enabled(False)


def _params(bgr2rgb=False, hw=None):
    c = cases.PRE_POST
    h, w = hw or c["image_hw"]
    return {"args": {"mean": c["mean"], "std": c["std"], "bgr2rgb": bgr2rgb, "resize_x": w, "resize_y": h}}


def test_oracle_matches_reference_fixtures():
    g, inp, c = golden("gv12_pre_post"), cases.pre_post_inputs(), cases.PRE_POST
    assert np.array_equal(g["standardized"], o_pp.standardize_rgb(inp["image_u8"], c["mean"], c["std"], False))
    logits = {k: torch.from_numpy(inp[k + "_logits"]) for k in ("static", "dynamic")}
    out = o_pp.post_process_train({"static_seg": logits["static"][:, None], "dynamic_seg": logits["dynamic"][:, None]})
    for k in ("static_map", "dynamic_map"):
        assert np.array_equal(g[k], out[k].numpy()), k
    for k in ("static_prob", "dynamic_prob"):
        assert np.abs(g[k] - out[k].numpy()).max() <= PROB_TOL, k
    assert np.array_equal(g["merged"], o_pp.merge_label(inp["road"], inp["lane"]))
    for i, (pred, gt) in enumerate(inp["pairs"]):
        assert np.array_equal(g["iu%d" % i], np.array(o_pp.mean_iu(pred, gt), dtype=np.float64))
        assert np.array_equal(g["precision%d" % i], np.array(o_pp.mean_precision(pred, gt), dtype=np.float64))
    got = o_pp.collate_batch(inp["samples"], train=True)["ego"]
    for k, v in got.items():
        assert str(v.dtype) == str(g["collate_dtype_" + k]) and np.array_equal(g["collate_" + k], v.numpy()), k


def test_rgb_preprocessor_host_mirror():
    g, inp = golden("gv12_pre_post"), cases.pre_post_inputs()
    pre = h_rgb.RgbPreProcessor(_params(), train=False)
    assert np.array_equal(g["standardized"], pre.preprocess(inp["image_u8"]))
    assert pre.preprocess(inp["image_u8"]).dtype == np.float64
    # BGR -> RGB is a channel reversal; standardisation is per channel so it commutes with it
    swapped = h_rgb.RgbPreProcessor(_params(bgr2rgb=True), train=False).preprocess(inp["image_u8"])
    c = cases.PRE_POST
    assert np.array_equal(swapped, o_pp.standardize_rgb(inp["image_u8"], c["mean"], c["std"], True))
    # cv2.resize (INTER_LINEAR) is restated from OpenCV's published algorithm - cv2 is absent, so PARITY UNPINNED: what can be
    # checked is the algorithm's own properties and an independent float evaluation of the same sampling positions
    img = inp["image_u8"]
    h, w = img.shape[:2]
    same = h_rgb.resize_linear(img, w, h)
    assert np.array_equal(same, img)                                          # identity size: taps (i, i+1) with weight 0
    const = np.full((11, 17, 3), 93, np.uint8)
    assert np.array_equal(h_rgb.resize_linear(const, 8, 6), np.full((6, 8, 3), 93, np.uint8))     # weights sum to 1 exactly
    rng = np.random.RandomState(0)
    big = rng.randint(0, 256, (24, 32, 3)).astype(np.uint8)
    half = h_rgb.resize_linear(big, 16, 12)                                   # exact 2x: OpenCV's 2x2-mean shortcut
    s = big.astype(np.int32)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    for (hh, ww) in ((6, 8), (17, 40), (30, 9)):
        got = h_rgb.resize_linear(big, ww, hh).astype(np.float64)
        ref = o_pp.resize_linear_float(big, ww, hh)
        assert got.shape == (hh, ww, 3) and np.abs(got - ref).max() <= 1.0 + 1e-9, (hh, ww)     # fixed-point vs float: <= 1 LSB
    out = h_rgb.RgbPreProcessor(_params(hw=(6, 8)), train=False).preprocess(img)
    assert out.shape == (6, 8, 3) and out.dtype == np.float64


def test_label_generation_host_mirror():
    g, inp = golden("gv12_pre_post"), cases.pre_post_inputs()
    post = h_post.CameraBevPostprocessor({}, train=False)
    assert np.array_equal(g["merged"], post.merge_label(inp["road"], inp["lane"]))
    lab = post.generate_label(inp["bev_bgr"])
    assert lab.dtype == np.float64 and np.array_equal(lab, o_pp.generate_label(inp["bev_bgr"]))
    # the crafted pixels around the gray > 0 threshold: B=4 -> 0, B=5 -> 1, (B,R)=(1,1) -> 0, (2,1) -> 1, G=1 -> 1
    assert lab[0, :6].tolist() == [0.0, 0.0, 1.0, 0.0, 1.0, 1.0]


def test_scores_host_arrays():
    g, inp = golden("gv12_pre_post"), cases.pre_post_inputs()
    for i, (pred, gt) in enumerate(inp["pairs"]):
        assert np.array_equal(g["iu%d" % i], np.array(h_seg.mean_IU(pred, gt), dtype=np.float64))
        assert np.array_equal(g["precision%d" % i], np.array(h_seg.mean_precision(pred, gt), dtype=np.float64))
    with pytest.raises(h_seg.EvalSegErr):
        h_seg.mean_IU(np.zeros((4, 5), dtype=np.int64), np.zeros((4, 6), dtype=np.int64))


def test_collate_batch_host_mirror():
    g, inp = golden("gv12_pre_post"), cases.pre_post_inputs()
    got = h_data.collate_batch(inp["samples"], train=True)["ego"]
    assert sorted(got.keys()) == sorted(k[len("collate_dtype_"):] for k in g.files if k.startswith("collate_dtype_"))
    for k, v in got.items():
        assert str(v.dtype) == str(g["collate_dtype_" + k]) and np.array_equal(g["collate_" + k], v.numpy()), k
    assert got["inputs"].shape == (5, 1, 4, 6, 8, 3) and got["record_len"].tolist() == [2, 3]
    with pytest.raises(AssertionError):
        h_data.collate_batch(inp["samples"], train=False)            # evaluation batches hold one scenario


def test_load_saved_model_host_mirror():
    g = golden("gv12_pre_post")
    with tempfile.TemporaryDirectory() as d:
        for ep in (3, 12, 7):
            torch.save({"weight": torch.full((2, 3), float(ep)), "stray.key": torch.zeros(1)}, os.path.join(d, "net_epoch%d.pth" % ep))
        ep, net = h_train.load_saved_model(d, nn.Linear(3, 2))
        assert ep == int(g["loaded_epoch"]) and np.array_equal(g["loaded_weight"], net.weight.detach().numpy())
    with tempfile.TemporaryDirectory() as d:
        assert h_train.load_saved_model(d, nn.Linear(3, 2))[0] == int(g["empty_epoch"]) == 0
    with pytest.raises(AssertionError):
        h_train.load_saved_model("/nonexistent/run/dir", nn.Linear(3, 2))


def test_vanilla_seg_loss_oracle_matches_reference():
    g = golden("gv14_vanilla_seg_loss")
    inp = {k: torch.from_numpy(v) for k, v in cases.seg_loss_inputs().items()}
    for i, args in enumerate(cases.SEG_LOSS):
        mine = o_pp.vanilla_seg_loss(args, inp, inp)
        for k in ("total_loss", "static_loss", "dynamic_loss"):
            assert abs(float(mine[k]) - float(g["%s%d" % (k, i)])) <= 1e-6 * max(1.0, abs(float(g["%s%d" % (k, i)])))


def _iou_updates(channels):
    return [(torch.from_numpy(p), {"bev": torch.from_numpy(b), "visibility": torch.from_numpy(v)}) for p, b, v in cases.iou_metric_inputs(channels)]


def test_nuscenes_iou_metric_oracle_matches_reference():
    g = golden("gv15_nuscenes_iou_metric")
    for i, c in enumerate(cases.IOU_METRIC):
        tp, fp, fn, res = o_pp.iou_metric(_iou_updates(c["channels"]), c["label_indices"], c["min_visibility"])
        assert np.array_equal(tp.numpy(), g["tp%d" % i]) and np.array_equal(fp.numpy(), g["fp%d" % i]) and np.array_equal(fn.numpy(), g["fn%d" % i])
        assert np.allclose([res[k] for k in sorted(res)], g["iou%d" % i], rtol=0, atol=1e-7)


def _focal_batch():
    inp = {k: torch.from_numpy(v) for k, v in cases.focal_loss_inputs().items()}
    return inp, {"bev": inp["bev"], "center": inp["center"], "visibility": inp["visibility"]}


def test_nuscenes_losses_oracle_matches_reference():
    g = golden("gv16_nuscenes_losses")
    inp, batch = _focal_batch()
    for i, c in enumerate(cases.FOCAL_LOSS):
        if c["kind"] == "bev":
            mine = o_pp.binary_segmentation_loss({"bev": inp["bev_pred"]}, batch, c["label_indices"], c["min_visibility"], c["alpha"], c["gamma"])
        else:
            mine = o_pp.center_loss({"center": inp["center_pred"]}, batch, c["min_visibility"], c["alpha"], c["gamma"])
        assert abs(float(mine) - float(g["loss%d" % i])) <= 1e-6 * max(1.0, float(g["loss%d" % i]))


def test_logit_side_has_no_cpu_fallback():
    post = h_post.CameraBevPostprocessor({}, train=False)
    with pytest.raises(CobevtHipError):
        post.softmax_argmax(torch.zeros(1, 2, 4, 4))


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_argmax_kernel(cuda, dtype):
    g, inp = golden("gv12_pre_post"), cases.pre_post_inputs()
    post = h_post.CameraBevPostprocessor({}, train=False)
    for name in ("static", "dynamic"):
        x = torch.from_numpy(inp[name + "_logits"])
        prob, seg = post.softmax_argmax(x.to(cuda).to(dtype))
        assert prob.dtype == torch.float32 and seg.dtype == torch.int64 and seg.shape == (x.shape[0],) + x.shape[2:]
        if dtype == torch.float32:
            ref_p, ref_m = g[name + "_prob"], g[name + "_map"]           # the reference's own outputs
        else:
            o_p, o_m = o_pp.softmax_argmax(x.to(dtype))                    # the oracle on the bf16-rounded logits
            ref_p, ref_m = o_p.numpy(), o_m.numpy()
        assert np.abs(prob.cpu().numpy() - ref_p).max() <= PROB_TOL
        got = seg.cpu().numpy()
        # the arg-max is taken over rounded probabilities: identical wherever the reference's top two differ by more than
        # the probability tolerance, and on every exact tie (first class wins)
        top2 = np.sort(ref_p, axis=1)[:, -2:]
        decided = (top2[:, 1] - top2[:, 0]) > 2 * PROB_TOL
        exact_tie = top2[:, 1] == top2[:, 0]
        assert np.array_equal(got[decided], ref_m[decided])
        assert np.array_equal(got[exact_tie], ref_m[exact_tie])
        assert (got != ref_m).sum() <= 1                                  # at most the crafted one-ulp pixel
        # the map IS the arg-max of the returned probabilities (first maximum)
        assert torch.equal(seg, torch.argmax(prob, dim=1))


@pytest.mark.gpu
def test_softmax_argmax_full_size_properties(cuda):
    """BASELINE-size head output (256 x 256, 2 and 3 classes): rows sum to 1, map == arg-max of prob, and the
    probabilities agree with the oracle."""
    for c in (2, 3):
        x = torch.randn(2, c, 256, 256, generator=torch.Generator().manual_seed(c)) * 4
        prob, seg = h_post.CameraBevPostprocessor({}, False).softmax_argmax(x.to(cuda))
        assert (prob.sum(1) - 1).abs().max().item() <= 1e-6
        assert torch.equal(seg, torch.argmax(prob, dim=1))
        o_p, o_m = o_pp.softmax_argmax(x)
        assert (prob.cpu() - o_p).abs().max().item() <= PROB_TOL
        assert (seg.cpu() != o_m).float().mean().item() <= 1e-5


@pytest.mark.gpu
def test_scores_on_device_maps(cuda):
    g, inp = golden("gv12_pre_post"), cases.pre_post_inputs()
    for i, (pred, gt) in enumerate(inp["pairs"]):
        p, t = torch.from_numpy(pred).to(cuda), torch.from_numpy(gt).to(cuda)
        assert np.array_equal(g["iu%d" % i], np.array(h_seg.mean_IU(p, t), dtype=np.float64))
        assert np.array_equal(g["precision%d" % i], np.array(h_seg.mean_precision(p, t), dtype=np.float64))
        assert np.array_equal(g["iu%d" % i], np.array(h_seg.mean_IU(p, gt), dtype=np.float64))     # host ground truth is fine
    # counts are exact at full size: compare with numpy bincount of the joint histogram
    from cobevt_amd import ops
    rs = np.random.RandomState(7)
    pred, gt = rs.randint(0, 3, size=(3, 256, 256)), rs.randint(0, 3, size=(3, 256, 256))
    counts = ops.seg_class_counts(torch.from_numpy(pred).to(cuda), torch.from_numpy(gt).to(cuda), 3).numpy()
    for n in range(3):
        for c in range(3):
            assert counts[n, c].tolist() == [int(((pred[n] == c) & (gt[n] == c)).sum()), int((gt[n] == c).sum()), int((pred[n] == c).sum())]
    with pytest.raises(CobevtHipError):
        ops.seg_class_counts(torch.full((1, 8, 8), 5, device=cuda), torch.zeros((1, 8, 8), dtype=torch.int64, device=cuda), 3)


@pytest.mark.gpu
def test_post_process_and_iou_end_to_end(cuda):
    """inference_camera.py:60-76: model output dict -> post_process -> cal_iou_training, device tensors throughout"""
    inp = cases.pre_post_inputs()
    sta, dyn = torch.from_numpy(inp["static_logits"]), torch.from_numpy(inp["dynamic_logits"])
    batch = h_data.collate_batch(inp["samples"], train=True)
    out = h_post.CameraBevPostprocessor({}, False).post_process(
        batch["ego"], {"static_seg": sta[:, None].to(cuda), "dynamic_seg": dyn[:, None].to(cuda)})
    iou_dynamic, iou_static = h_seg.cal_iou_training(batch, out)
    ref = o_pp.post_process_train({"static_seg": sta[:, None], "dynamic_seg": dyn[:, None]})
    # score the oracle's scores on OUR maps (they may differ from the oracle's on the one crafted near-tie pixel)
    ref_dyn = o_pp.mean_iu(out["dynamic_map"][0].cpu().numpy(), batch["ego"]["gt_dynamic"][0, 0].numpy())
    ref_sta = o_pp.mean_iu(out["static_map"][0].cpu().numpy(), batch["ego"]["gt_static"][0, 0].numpy())
    assert iou_dynamic == ref_dyn and iou_static == ref_sta
    assert (out["static_map"].cpu() != ref["static_map"]).sum().item() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vanilla_seg_loss_forward(cuda, dtype):
    """validation loss of train_camera.py:182-196 on device logits vs the reference's own values (gv14); bf16 logits vs the
    oracle on the rounded logits.  1e-5 rel: a 1680-term fp32 sum in a different order and a different libm."""
    from cobevt_amd.host.vanilla_seg_loss import VanillaSegLoss
    g = golden("gv14_vanilla_seg_loss")
    inp = {k: torch.from_numpy(v) for k, v in cases.seg_loss_inputs().items()}
    for i, args in enumerate(cases.SEG_LOSS):
        crit = VanillaSegLoss(dict(args))
        out = {"static_seg": inp["static_seg"].to(cuda).to(dtype), "dynamic_seg": inp["dynamic_seg"].to(cuda).to(dtype)}
        gt = {"gt_static": inp["gt_static"].to(cuda), "gt_dynamic": inp["gt_dynamic"]}            # host ground truth is accepted
        total = crit(out, gt)
        assert total.is_cuda and float(total) == float(crit.loss_dict["total_loss"])
        if dtype == torch.float32:
            ref = {k: float(g["%s%d" % (k, i)]) for k in ("total_loss", "static_loss", "dynamic_loss")}
        else:
            ref = {k: float(v) for k, v in o_pp.vanilla_seg_loss(args, {k2: v2.cpu().float() for k2, v2 in out.items()}, inp).items()}
        for k, r in ref.items():
            assert abs(float(crit.loss_dict[k]) - r) <= 1e-5 * max(1.0, abs(r)), (k, args["target"], float(crit.loss_dict[k]), r)
    # full-size head output: equals the oracle, and is bit-reproducible (fixed summation order)
    from cobevt_amd import ops
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(3)) * 3
    y = torch.randint(0, 3, (2, 256, 256), generator=torch.Generator().manual_seed(4))
    wt = torch.tensor([1.0, 2.0, 4.0])
    a = ops.weighted_cross_entropy(x.to(cuda), y.to(cuda), wt)
    assert torch.equal(a, ops.weighted_cross_entropy(x.to(cuda), y.to(cuda), wt))
    ref = torch.nn.functional.cross_entropy(x, y, weight=wt)
    assert abs(float(a) - float(ref)) <= 1e-5 * float(ref)
    # ignore_index -100 drops the pixel from numerator and denominator like nn.CrossEntropyLoss; any other label outside
    # [0, C) raises as the reference does - one call late or at check_deferred_label_errors(): the count leaves the device without a
    # host sync (a blocking read drained the launch queue twice per training step)
    y2 = y.clone()
    y2[0, :7, :9] = -100
    a2 = ops.weighted_cross_entropy(x.to(cuda), y2.to(cuda), wt)
    ref2 = torch.nn.functional.cross_entropy(x, y2, weight=wt)
    assert abs(float(a2) - float(ref2)) <= 1e-5 * float(ref2)
    y3 = y.clone()
    y3[1, 200, 100] = 255
    ops.check_deferred_label_errors()                       # nothing pending from the calls above
    ops.weighted_cross_entropy(x.to(cuda), y3.to(cuda), wt)  # does not sync, does not raise yet
    with pytest.raises(Exception, match="outside"):
        ops.check_deferred_label_errors()
    ops.check_deferred_label_errors()                       # reported once


@pytest.mark.gpu
def test_nuscenes_iou_metric_on_device(cuda):
    """IoUMetric.update / compute with the prediction on the GPU: integer tp / fp / fn equal to the reference's (gv15) -
    including the pixels crafted to sit exactly on the 0.5 threshold; the three next to a threshold may flip with the exp
    implementation, so the counts are allowed to differ by that many."""
    from cobevt_amd.host.nuscenes.metrics import BaseIoUMetric, IoUMetric
    g = golden("gv15_nuscenes_iou_metric")
    for i, c in enumerate(cases.IOU_METRIC):
        m = IoUMetric(c["label_indices"], c["min_visibility"])
        for pred, batch in _iou_updates(c["channels"]):
            m.update({"bev": pred.to(cuda)}, {k: v.to(cuda) for k, v in batch.items()})
        for name in ("tp", "fp", "fn"):
            assert np.abs(getattr(m, name).numpy() - g["%s%d" % (name, i)]).max() <= 3, name
        res = m.compute()
        assert sorted(res) == ["@0.40", "@0.50"]
        assert np.allclose([res[k] for k in sorted(res)], g["iou%d" % i], rtol=0, atol=2e-3)
        m.reset()
        assert float(m.tp.sum()) == 0
    # BaseIoUMetric.update(pred, label) on flat tensors, vs the oracle on the same data
    pred, batch = _iou_updates(1)[0]
    label = batch["bev"][:, 4:5]
    base = BaseIoUMetric()
    base.update(pred.to(cuda), label.to(cuda))
    p = pred.sigmoid().reshape(-1)[:, None] >= base.thresholds[None]
    l = label.bool().reshape(-1)[:, None]
    assert np.abs(base.tp.numpy() - (p & l).sum(0).numpy()).max() <= 3 and np.abs(base.fn.numpy() - (~p & l).sum(0).numpy()).max() <= 3


@pytest.mark.gpu
def test_nuscenes_losses_forward_on_device(cuda):
    """BinarySegmentationLoss / CenterLoss / MultipleLoss forward with the predictions on the GPU vs the reference's values
    (gv16; fvcore's focal loss restated).  1e-5 rel: fp32 sums of ~4000 terms in a different order, different libm."""
    from cobevt_amd.host.nuscenes.losses import BinarySegmentationLoss, CenterLoss, MultipleLoss, SigmoidFocalLoss
    g = golden("gv16_nuscenes_losses")
    inp, batch = _focal_batch()
    dbatch = {k: v.to(cuda) for k, v in batch.items()}
    pred = {"bev": inp["bev_pred"].to(cuda), "center": inp["center_pred"].to(cuda)}
    for i, c in enumerate(cases.FOCAL_LOSS):
        if c["kind"] == "bev":
            val = BinarySegmentationLoss(c["label_indices"], c["min_visibility"], c["alpha"], c["gamma"])(pred, dbatch)
        else:
            val = CenterLoss(c["min_visibility"], c["alpha"], c["gamma"])(pred, dbatch)
        ref = float(g["loss%d" % i])
        assert val.is_cuda and abs(float(val) - ref) <= 1e-5 * max(1.0, ref), (c, float(val), ref)
    total, parts = MultipleLoss({"bev": BinarySegmentationLoss([[4, 5]], 2), "bev_weight": 1.0,
                                 "center": CenterLoss(2), "center_weight": 0.1})(pred, dbatch)
    assert sorted(parts) == ["bev", "center"] and abs(float(total) - float(g["multi_total"])) <= 1e-5 * max(1.0, float(g["multi_total"]))
    # plain SigmoidFocalLoss(pred, label) == the oracle's element-wise definition, mean
    x, t = inp["center_pred"], inp["center"]
    val = SigmoidFocalLoss(alpha=0.25, gamma=2.0)(x.to(cuda), t.to(cuda))
    ref = float(o_pp.sigmoid_focal_loss(x, t, 0.25, 2.0).mean())
    assert abs(float(val) - ref) <= 1e-5 * max(1.0, ref)
