"""CPU: drop-in boundary checks that need no GPU — state_dict schema, registry, C-ABI symbols, fail-loud."""
import copy
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from cobevt_amd import host, lib, synth
from cobevt_amd.lib import CobevtHipError
from cobevt_amd import registry
from cobevt_amd.registry import create_model
from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,cfg", [("corpbevt_full", synth.corpbevt_config()), ("corpbevt_small", synth.corpbevt_small_config())])
def test_state_dict_schema_matches_reference(name, cfg):
    g = golden("gv0_state_dict_schema")
    ref = dict(zip(g[name + "_keys"].tolist(), g[name + "_shapes"].tolist()))
    m = host.CorpBEVT(copy.deepcopy(cfg))
    mine = {k: ",".join(str(int(d)) for d in v.shape) for k, v in m.state_dict().items()}
    missing = sorted(set(ref) - set(mine))
    assert not missing, "reference keys absent: %s" % missing[:5]
    # identical key sets, incl. the unused torchvision `fc` the encoder keeps (resnet_ms.py:38, Appendix D)
    assert set(mine) == set(ref) and "encoder.encoder.fc.weight" in mine
    bad = [k for k in ref if ref[k] != mine[k]]
    assert not bad, "shape mismatch: %s" % [(k, ref[k], mine[k]) for k in bad[:5]]
    # a reference checkpoint (plain state_dict) loads the way the reference loads it (strict=False, train_utils.py:24-65)
    res = m.load_state_dict({k: torch.zeros([int(d) for d in s.split(",")] if s else []) for k, s in ref.items()}, strict=False)
    assert not res.unexpected_keys


def test_registry_contract():
    cfg = synth.corpbevt_small_config()
    m = create_model({"model": {"core_method": "corpbevt", "args": copy.deepcopy(cfg)}})
    assert type(m).__name__ == "CorpBEVT"
    cfg2 = {k: copy.deepcopy(v) for k, v in cfg.items() if k in ("target", "encoder", "decoder", "fax", "seg_head_dim", "output_class")}
    m2 = create_model({"model": {"core_method": "fax_fused_transformer", "args": cfg2}})
    assert type(m2).__name__ == "FaxFusedTransformer"
    with pytest.raises(ValueError):
        create_model({"model": {"core_method": "no_such_model", "args": {}}})


def test_loss_registry_contract():
    from cobevt_amd.registry import create_loss
    crit = create_loss({"loss": {"core_method": "vanilla_seg_loss",
                                 "args": {"d_weights": 75.0, "s_weights": 15.0, "d_coe": 2.0, "s_coe": 0.0, "target": "dynamic"}}})
    assert type(crit).__name__ == "VanillaSegLoss" and crit.l_weights == 50 and crit.target == "dynamic"
    with pytest.raises(ValueError):
        create_loss({"loss": {"core_method": "no_such_loss", "args": {}}})


def test_c_abi_exports_every_declared_symbol():
    """the shared library loads and exports exactly what include/cobevt_hip.h declares (no compute here)."""
    header = open(os.path.join(ROOT, "include", "cobevt_hip.h")).read()
    declared = set(re.findall(r"\b(cobevt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.SIGNATURES), "ctypes table and header disagree: %s" % (declared ^ set(lib.SIGNATURES))
    l = lib.load()
    for name in declared:
        assert getattr(l, name) is not None
    assert l.cobevt_abi_version() == 1
    assert l.cobevt_strerror(2).decode() == "unsupported shape / alignment"
    # argument validation happens before any launch: null pointers are rejected without touching a device
    dims = (ctypes.c_int * 21)(*([0] * 21))
    assert l.cobevt_conv2d_nhwc(None, None, None, None, None, None, None, None, dims, None) == 1


def test_second_library_and_compute_modes():
    """libcobevt_hip_f32s.so (same sources, -DCOBEVT_F32_SPLIT=1: fp32 storage on the split-bf16 matrix path) loads, exports the whole
    C ABI, and host.set_compute_dtype selects it - "fp32_split" is fp32 storage, so the lowered plans are shared with exact fp32"""
    l2 = lib.load("f32s")
    assert l2 is not lib.load("") and l2.cobevt_abi_version() == 1
    for name in lib.SIGNATURES:
        assert getattr(l2, name) is not None
    assert lib.get_variant() == "" and host.get_compute_mode() == "bf16"
    with host.compute_dtype("fp32_split"):
        assert host.get_compute_dtype() == torch.float32 and host.get_matrix_path() == "split_bf16" and host.get_compute_mode() == "fp32_split"
        assert lib.get_variant() == "f32s" and lib.load() is l2
        with host.compute_dtype(torch.bfloat16):
            assert lib.get_variant() == "" and host.get_compute_mode() == "bf16"
        assert lib.get_variant() == "f32s"
    assert lib.get_variant() == "" and host.get_compute_mode() == "bf16"
    with host.compute_dtype(torch.float32):
        assert host.get_compute_mode() == "fp32" and lib.get_variant() == ""
    with pytest.raises(CobevtHipError):
        host.set_compute_dtype(torch.bfloat16, "split_bf16")
    with pytest.raises(CobevtHipError):
        host.set_compute_dtype("fp64")


def test_third_library_is_the_encoders_library_in_fast_mode():
    """libcobevt_hip_f32h.so (-DCOBEVT_F32_SPLIT=2: one fp16 MFMA per piece, weights as a single fp16 term) loads and exports the whole
    C ABI; "fp32_fast" = the fp32_split library everywhere, with the launches issued inside lib.encoder_scope() - the ResNet encoder's,
    host/resnet_ms.py - going to the third library; no other mode has an encoder override."""
    l3 = lib.load("f32h")
    assert l3 is not lib.load("") and l3 is not lib.load("f32s") and l3.cobevt_abi_version() == 1
    for name in lib.SIGNATURES:
        assert getattr(l3, name) is not None
    assert lib.get_encoder_variant() is None
    with lib.encoder_scope():
        assert lib.get_variant() == ""                     # no override outside fp32_fast
    with host.compute_dtype("fp32_fast"):
        assert host.get_compute_dtype() == torch.float32 and host.get_compute_mode() == "fp32_fast"
        assert host.get_matrix_path() == "split_bf16_enc_fp16"
        assert lib.get_variant() == "f32s" and lib.get_encoder_variant() == "f32h"
        with lib.encoder_scope():
            assert lib.get_variant() == "f32h" and lib.load() is l3
            with lib.encoder_scope():                      # re-entrant
                assert lib.get_variant() == "f32h"
            assert lib.get_variant() == "f32h"
        assert lib.get_variant() == "f32s"
        with host.compute_dtype("fp32_split"):
            assert lib.get_encoder_variant() is None
        assert lib.get_encoder_variant() == "f32h"
    assert lib.get_variant() == "" and lib.get_encoder_variant() is None and host.get_compute_mode() == "bf16"
    with pytest.raises(CobevtHipError):
        host.set_compute_dtype(torch.bfloat16, "split_bf16_enc_fp16")


def test_uint8_ingest_table_is_the_preprocessors_arithmetic():
    """the (3, 256) table the stem kernel reads == RgbPreProcessor.standalize(normalize(.)) + the collate cast, value for value"""
    from cobevt_amd.host.rgb_preprocessor import RgbPreProcessor, normalisation_table
    args = {"bgr2rgb": True, "resize_x": 16, "resize_y": 16, "mean": list(synth.OPV2V_RGB_MEAN), "std": list(synth.OPV2V_RGB_STD)}
    pre = RgbPreProcessor({"args": args}, train=False)
    frame = np.arange(256, dtype=np.uint8).repeat(3).reshape(16, 16, 3)
    want = pre.standalize(pre.normalize(frame)).astype(np.float32)                # (16, 16, 3), byte value u at flat pixel u
    t = pre.normalisation_table()
    assert t.dtype == np.float32 and t.shape == (3, 256)
    for c in range(3):
        assert np.array_equal(t[c], want.reshape(256, 3)[:, c])
    assert np.array_equal(t, normalisation_table(args["mean"], args["std"]))
    enc = host.ResnetEncoder({"num_layers": 18, "pretrained": False, "image_height": 64, "image_width": 64, "id_pick": [1]})
    keys = set(enc.state_dict())
    enc.set_rgb_normalisation(args["mean"], args["std"], bgr2rgb=True)
    assert set(enc.state_dict()) == keys and torch.equal(enc.ingest_lut, torch.from_numpy(t).flip(0)) and enc.ingest_bgr
    b8, b32 = synth.opv2v_batch_u8(1, cams=1, image=16, max_cav=2, seed=3, bgr=True)
    rgb = b8["inputs"].numpy()[..., ::-1]
    assert np.array_equal(b32["inputs"].numpy(), pre.standalize(pre.normalize(rgb)).astype(np.float32))


def test_no_cpu_fallback_and_eval_only():
    m = host.FaxAttention(64, 32, 0.0, 8)
    x = torch.zeros(1, 64, 8, 8)
    with pytest.raises(CobevtHipError):           # training mode is rejected
        m(x)
    with pytest.raises(CobevtHipError):           # CPU tensors are rejected: the product path never falls back
        m.eval()(x)
    with pytest.raises(ValueError):
        host.ResnetEncoder({"num_layers": 19, "pretrained": False, "image_height": 64, "image_width": 64, "id_pick": [1]})
    m = host.CorpBEVT(dict(synth.corpbevt_small_config(), compression=4))       # corpbevt.py:79-81: NaiveCompressor(128, ratio)
    assert m.compression and tuple(m.naive_compressor.encoder[0].weight.shape) == (32, 128, 3, 3)
    assert not host.CorpBEVT(synth.corpbevt_small_config()).compression


def test_product_does_not_import_oracle():
    """cobevt_amd/ must never import oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "cobevt_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), "%s imports oracle" % f


def test_procedural_weights_are_reproducible():
    a = synth.procedural_tensor("fax.cross_views.0.mlp_1.0.weight", (256, 128), 0)
    b = synth.procedural_tensor("fax.cross_views.0.mlp_1.0.weight", (256, 128), 0)
    assert torch.equal(a, b) and abs(a.mean().item()) < 0.01
    # pinned values: a silent change of the generator would invalidate every fixture
    assert np.allclose(a.flatten()[:3].numpy(), synth.procedural_tensor("fax.cross_views.0.mlp_1.0.weight", (256, 128)).flatten()[:3].numpy())
    assert synth.procedural_tensor("x.num_batches_tracked", ()) is None


def test_kernel_plans_follow_the_parameters():
    """a module's cached kernel plans (folded / re-laid-out weights) are rebuilt when its parameters change - in-place
    updates and load_state_dict of a reference checkpoint (train_utils.py:54-63) - and are per compute dtype"""
    from cobevt_amd.host import runtime as rt
    m = host.NaiveCompressor(16, 2).eval()
    conv, bn = m.encoder[0], m.encoder[1]
    with host.compute_dtype(torch.float32):
        p1 = rt.conv_plan(m, "enc", conv, bn, act=1)
        assert rt.conv_plan(m, "enc", conv, bn, act=1) is p1                    # cached
        with torch.no_grad():
            conv.weight.mul_(2.0)                                               # in-place update bumps the version
        p2 = rt.conv_plan(m, "enc", conv, bn, act=1)
        assert p2 is not p1 and torch.allclose(p2.wgt, 2 * p1.wgt)
        sd = {k: (v * 0 + 1 if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        p3 = rt.conv_plan(m, "enc", conv, bn, act=1)
        assert p3 is not p2 and float(p3.wgt.abs().max()) > 0 and not torch.allclose(p3.wgt, p2.wgt)
    with host.compute_dtype(torch.bfloat16):
        p4 = rt.conv_plan(m, "enc", conv, bn, act=1)
        assert p4 is not p3 and p4.wgt.dtype == torch.bfloat16


def test_registry_reports_real_import_errors(tmp_path, monkeypatch):
    """A module that exists but fails to import must surface its own error, not 'backbone not found'."""
    import cobevt_amd.host as hostpkg
    broken = os.path.join(os.path.dirname(hostpkg.__file__), "zz_broken_for_test.py")
    with open(broken, "w") as f:
        f.write("import a_dependency_that_does_not_exist_xyz\n")
    try:
        with pytest.raises(ModuleNotFoundError, match="a_dependency_that_does_not_exist_xyz"):
            registry.create_model({"model": {"core_method": "zz_broken_for_test", "args": {}}})
    finally:
        os.remove(broken)
    with pytest.raises(ValueError, match="not found"):
        registry.create_model({"model": {"core_method": "no_such_model", "args": {}}})


def test_invalidate_plans_sees_dot_data_writes():
    """Plan caches are keyed on the parameter version; `.data` writes bypass it -> invalidate_plans() is the documented hook,
    and load_state_dict() calls it."""
    from cobevt_amd.host.runtime import HipModule

    class M(HipModule):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(4, 4)
            self.child = None

    m, c = M(), M()
    m.child = c
    built = []
    for mod in (m, c):
        mod._plan("p", [mod.lin.weight], lambda dt, dev: built.append(1) or object())
    m.lin.weight.data.mul_(2.0)                                   # invisible to the version counter
    m._plan("p", [m.lin.weight], lambda dt, dev: built.append(1) or object())
    assert len(built) == 2
    m.invalidate_plans()
    assert not m._plan_cache and not c._plan_cache
    m._plan("p", [m.lin.weight], lambda dt, dev: built.append(1) or object())
    assert len(built) == 3
    m.load_state_dict(m.state_dict())
    assert not m._plan_cache


def test_driver_entry_points_compile_and_exist():
    """__graft_entry__.build / smoke and bench.py are what the driver runs unattended: they must at least compile and import on a
    machine without a GPU (a stray syntax error there fails the whole round silently for every other test)"""
    import importlib
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("__graft_entry__.py", "bench.py"):
        py_compile.compile(os.path.join(root, name), doraise=True)
    for name in sorted(os.listdir(os.path.join(root, "tools"))):
        if name.endswith(".py"):
            py_compile.compile(os.path.join(root, "tools", name), doraise=True)
    mod = importlib.import_module("__graft_entry__")
    assert callable(mod.build) and callable(mod.smoke)


def test_training_host_index_helpers():
    """host/training.py: the (sample, slot) of every un-grouped agent row from record_len without a host round trip, and the choice of the
    blocked weight-gradient form per convolution shape"""
    import torch
    from cobevt_amd import autograd as ag
    from cobevt_amd.host import training
    rl = torch.tensor([2, 1, 3], dtype=torch.int32)
    b_of, i_of = training._agent_index(rl, 6)
    assert b_of.tolist() == [0, 0, 1, 2, 2, 2] and i_of.tolist() == [0, 1, 0, 0, 1, 2]
    assert ag.wgrad_blocked_mode(3, 1, 1, 64) == 0 and ag.wgrad_blocked_mode(1, 1, 0, 128) == 0
    assert ag.wgrad_blocked_mode(3, 2, 1, 64) == 1 and ag.wgrad_blocked_mode(1, 2, 0, 64) == 1
    assert ag.wgrad_blocked_mode(7, 2, 3, 3) == 2                       # the stem: 7 tap columns x 3 channels = 21 pseudo-channels
    assert ag.wgrad_blocked_mode(7, 2, 3, 8) is None                     # 56 pseudo-channels do not fit one 32-lane operand
    assert ag.wgrad_blocked_mode(3, 2, 0, 64) is None and ag.wgrad_blocked_mode(5, 1, 2, 64) is None
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    wt = training._flipped_taps(w)
    assert all(float(wt[o, c, u, v]) == float(w[o, c, v, 2 - u]) for o in range(2) for c in range(3) for u in range(3) for v in range(3))


def test_bench_line_finishing_touches():
    """bench.finish_line (pure host logic): top-level copies of the other configs' figures, fractions against the box calibration
    beside the spec-peak ones, the headline definition - and nothing breaks when legs are missing or carry an error"""
    import bench
    line = {"value": 600.0, "unit": "frames/s", "ms_per_step": 1.66, "ms_per_step_median": 1.66, "scaling": "weak", "mode": "throughput",
            "dtype": "bf16", "config": {"workload": "OPV2V-camera CoBEVT (corpbevt.yaml): 5 agents"},
            "box_calibration": {"mfma_bf16_tflops": 2000.0, "hbm_copy_gbs": 5000.0, "sclk_mhz_under_mfma_load": 2000.0},
            "roofline": {"kernel": "conv3x3", "bound": "mfma", "achieved": 600.0, "peak": 2500.0, "frac": 0.24},
            "roofline_other_kernels": [{"kernel": "gemm_rows", "bound": "hbm", "achieved": 1000.0, "frac": 0.125},
                                       {"kernel": "row_chain", "bound": "valu", "achieved": 128.0, "frac": 0.2083}, {"error": "x"}],
            "roofline_fax_attention": {"bound": "mfma", "achieved": 500.0, "frac": 0.2},
            "fp32_parity_mode": {"frames_per_sec": 140.0}, "one_frame_at_a_time": {"error": "boom"},
            "other_configs": {"lidar_fusebevt": {"frames_per_sec": 515.0, "roofline": [{"kernel": "attention", "bound": "mfma", "achieved": 400.0}]},
                              "nuscenes_sinbevt_from_images": {"error": "no"}, "train_step_bf16_autocast": {"frames_per_sec": 40.0}}}
    bench.finish_line(line, 1)
    assert line["lidar_fusebevt_frames_per_sec"] == 515.0 and line["fp32_parity_mode_frames_per_sec"] == 140.0
    assert line["train_steps_per_sec_bf16_autocast"] == 40.0
    assert "nuscenes_sinbevt_from_images_frames_per_sec" not in line and "one_frame_at_a_time_frames_per_sec" not in line
    assert line["roofline"]["frac_of_box_calibrated_peak"] == 0.3
    assert line["roofline_other_kernels"][0]["frac_of_box_calibrated_peak"] == 0.2
    assert line["roofline_other_kernels"][1]["frac_of_box_calibrated_peak"] == 0.25          # 128 G wave-instr/s of 1024 x 2.0 GHz / 4
    assert line["roofline_fax_attention"]["useful_mfma_frac_of_box_calibrated_peak"] == 0.25
    assert line["other_configs"]["lidar_fusebevt"]["roofline"][0]["frac_of_box_calibrated_peak"] == 0.2
    assert line["headline_version"] == 2 and "throughput" in line["headline_definition"] and "throughput_mode" not in line
    multi = dict(line, n_gpus=8)
    bench.finish_line(multi, 8)
    assert multi["throughput_mode"]["value"] == 600.0 and multi["throughput_mode"]["mode"] == "throughput"
    bench.finish_line({"value": 1.0, "config": {"workload": "OPV2V-LiDAR FuseBEVT"}}, 1)     # a line without any of the legs


def test_host_frame_feeder_protocol(monkeypatch):
    """HostFrameFeeder's hand-over protocol on a stand-in runner (no GPU: the fetch kernel and the events are recorded, not run): which
    pinned slot is fetched into which image slot and when, which event a put() waits for before it rewrites a ring slot, and that a frame
    handed over AFTER the previous step (put, step, put, step) is fetched in stream order instead of being lost"""
    from cobevt_amd.host import pipeline
    log = []

    class Event(object):
        count = 0

        def __init__(self):
            Event.count += 1
            self.id = Event.count

        def record(self):
            log.append(("record", self.id))

        def synchronize(self):
            log.append(("wait", self.id))

    def runner():
        r = object.__new__(pipeline.PipelinedCorpBEVT)
        r.host_ingest, r.depth, r.i = True, 3, 0
        r.pinned = [torch.zeros(16, dtype=torch.uint8) for _ in range(3)]
        r.slots = [{"inputs": torch.zeros(16, dtype=torch.uint8), "intrinsic": None, "extrinsic": None, "transformation_matrix": None,
                    "record_len": None} for _ in range(3)]

        def step(small):
            assert small["inputs"] is r.slots[r.i % 3]["inputs"]
            log.append(("step", r.i))
            r.i += 1
            return r.i - 1
        r.step = step
        return r

    def name(r, t):
        for i in range(3):
            if t.data_ptr() == r.pinned[i].data_ptr():
                return "pinned%d" % i
            if t.data_ptr() == r.slots[i]["inputs"].data_ptr():
                return "slot%d" % i
        return "?"
    monkeypatch.setattr(pipeline.torch.cuda, "Event", Event)
    small = {"intrinsic": 1, "extrinsic": 2, "transformation_matrix": 3, "record_len": 4}
    frame = lambda: dict(small, inputs=torch.ones(16, dtype=torch.uint8))          # noqa: E731

    # one frame ahead of the step: only the very first frame is fetched outside a graph; put(k + 3) waits for the read of frame k
    r = runner()
    monkeypatch.setattr(pipeline.ops, "host_fetch", lambda src, dst, blocks=0: log.append(("fetch", name(r, src), name(r, dst))))
    f = pipeline.HostFrameFeeder(r)
    f.put(frame())
    first = [e for e in log if e[0] == "record"][0][1]
    for k in range(5):
        f.put(frame())
        assert f.step() == k
    assert [e for e in log if e[0] == "fetch"] == [("fetch", "pinned0", "slot0")]
    waits = [e[1] for e in log if e[0] == "wait"]
    records = [e[1] for e in log if e[0] == "record"]
    assert waits[0] == first and waits[1:] == records[1:1 + len(waits) - 1]     # put(3) waits for the first fetch, put(4) for step 0's pull, ..
    with pytest.raises(CobevtHipError):
        f.put(frame()); f.put(frame())

    # put AFTER the step: every frame after the first missed the pull and is fetched (pinned q -> image slot q) before its own step
    del log[:]
    r = runner()
    f = pipeline.HostFrameFeeder(r)
    for k in range(5):
        f.put(frame())
        assert f.step() == k
    order = [e for e in log if e[0] in ("fetch", "step")]
    want = [("fetch", "pinned0", "slot0"), ("step", 0)]
    for k in range(1, 5):
        want += [("fetch", "pinned%d" % (k % 3), "slot%d" % (k % 3)), ("step", k)]
    assert order == want
    with pytest.raises(CobevtHipError):
        f.step()                                                   # nothing handed over
    with pytest.raises(CobevtHipError):
        f.put(dict(small, inputs=torch.ones(8, dtype=torch.uint8)))

    # the small tensors are COPIED at put(): a loader that reuses its pose buffers for frame k + 1 before step(k) must not
    # change what step(k) uploads; a set of the feeder's ring is rewritten only behind the event of the step that uploaded it
    del log[:]
    r = runner()
    seen = []
    inner = r.step
    r.step = lambda sm: (seen.append(float(sm["transformation_matrix"][0])), inner(sm))[1]
    f = pipeline.HostFrameFeeder(r)
    pose = torch.zeros(4)
    mk = lambda: {"inputs": torch.ones(16, dtype=torch.uint8), "intrinsic": 1, "extrinsic": 2, "transformation_matrix": pose, "record_len": 4}   # noqa: E731
    pose.fill_(0.0)
    f.put(mk())
    for k in range(8):
        pose.fill_(float(k + 1))              # the loader's buffer, rewritten for frame k + 1 ...
        f.put(mk())
        assert f.step() == k                  # ... before frame k is stepped
    assert seen == [float(k) for k in range(8)]
    n = len(f.small_sets)
    assert n == 6 and f.small_sets[0]["transformation_matrix"].data_ptr() != pose.data_ptr()
    step_events = [e[1] for e in log if e[0] == "record"][1:]           # [0] = the first frame's fetch
    late_waits = [e[1] for e in log if e[0] == "wait"]
    assert step_events[0] in late_waits and step_events[1] in late_waits  # puts 6, 7 reused sets 0, 1: behind the events of steps 0, 1


def test_plan_fingerprint_follows_replaced_tensors():
    """AgentCountPlans lists the model's tensors once and re-lists them when the structure epoch moves; every way a module
    can REPLACE a tensor object must move it (torch's global registration hooks, host/runtime.py), or a captured graph would keep
    replaying the old weights with an unchanged fingerprint"""
    import torch.nn as nn
    from cobevt_amd.host import pipeline, runtime as rt

    class Tiny(rt.HipModule):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)
            self.bn = nn.BatchNorm1d(4)
    m = Tiny()
    plans = pipeline.AgentCountPlans(m, use_graph=False)
    fp0 = plans.weights_fingerprint()
    assert plans.weights_fingerprint() == fp0
    m.lin.weight = nn.Parameter(torch.ones(4, 4))                   # a new Parameter OBJECT on a plain nn.Linear container
    fp1 = plans.weights_fingerprint()
    assert fp1 != fp0 and any(t is m.lin.weight for t in plans._tensors)
    m.bn.running_mean = torch.full((4,), 2.0)                        # a registered buffer re-assigned
    fp2 = plans.weights_fingerprint()
    assert fp2 != fp1 and any(t is m.bn.running_mean for t in plans._tensors)
    m.bn.register_buffer("extra_scale", torch.ones(4))               # a tensor added
    assert len(plans.weights_fingerprint()) == len(fp2) + 1
    m.lin = nn.Linear(4, 4)                                          # a child module replaced (BN fusing, surgery)
    fp3 = plans.weights_fingerprint()
    assert any(t is m.lin.weight for t in plans._tensors)
    with torch.no_grad():
        m.lin.weight.mul_(2.0)                                       # in place: seen through the version counter, no re-listing needed
    assert plans.weights_fingerprint() != fp3
    e = rt.structure_epoch()
    m.float()                                                        # conversions re-list as well (HipModule._apply)
    assert rt.structure_epoch() > e


def test_projection_shortcut_fragment_table_layout():
    """ops._conv3_ds_table (host side of cobevt_conv3x3_ds_wfrag_nhwc): per 32-cout tile the 3x3's Cin/64 * 9 fragment steps followed by
    the shortcut's Cin2/64 one-tap steps.  Decoded back on the CPU exactly as the kernel addresses it - step, k-group g, lane = 32 * half
    + cout % 32, eight consecutive channels 64 * chunk + 16 * g + 8 * half .. - the table reproduces conv3x3(y; W2) + conv1x1(x; Wd) of the
    folded weights (no GPU: only the layout the kernel is handed)."""
    from cobevt_amd import ops
    g = torch.Generator().manual_seed(3)
    c, cin2 = 128, 64
    w2 = torch.randn(c, c, 3, 3, generator=g) * 0.05
    wd = torch.randn(c, cin2, 1, 1, generator=g) * 0.1
    p2 = ops.ConvPlan(w2, None, stride=1, pad=1, act=1, dtype=torch.bfloat16, device="cpu")
    pd = ops.ConvPlan(wd, None, stride=2, pad=0, act=0, dtype=torch.bfloat16, device="cpu")
    table = ops._conv3_ds_table(p2, pd).float()
    assert ops._conv3_ds_table(p2, pd) is p2._ds_fused[1]                      # cached per shortcut plan
    tiles, nmain, nextra = p2.coutp3 // 32, (c // 64) * 9, cin2 // 64
    assert table.shape == (tiles, nmain + nextra, 2048)
    t = table.reshape(tiles, nmain + nextra, 4, 2, 32, 8)                      # [tile][step][k-group][half][cout % 32][8 channels]
    w2r, wdr = w2.to(torch.bfloat16).float(), wd.to(torch.bfloat16).float()
    for cout in (0, 37, 127):
        tile, r = cout // 32, cout % 32
        for chunk in range(c // 64):
            for tap in range(9):
                got = t[tile, chunk * 9 + tap, :, :, r, :].reshape(64)          # channel 16 g + 8 half + e of the chunk
                assert torch.equal(got, w2r[cout, chunk * 64:(chunk + 1) * 64, tap // 3, tap % 3])
        for chunk in range(nextra):
            got = t[tile, nmain + chunk, :, :, r, :].reshape(64)
            assert torch.equal(got, wdr[cout, chunk * 64:(chunk + 1) * 64, 0, 0])
