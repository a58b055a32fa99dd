"""GPU: the package-level HIP-graph runners (cobevt_amd/host/pipeline.py) reproduce `model(batch)` bit for bit - one frame at a
time (CapturedCorpBEVT) and with several frames in flight (PipelinedCorpBEVT, depth 3 and 4), fed a DIFFERENT frame every
step so that a stale ring slot, a wrong pose slot or a missed dependency between the streams shows up as a mismatch."""
import copy

import pytest
import torch

import cases
from cobevt_amd import host, synth
from cobevt_amd.host import pipeline
from cobevt_amd.lib import CobevtHipError
from cobevt_amd.synth import fill_module_

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _frames(n, cuda, agents=2):
    out = []
    for f in range(n):
        b = synth.opv2v_batch(agents=agents, cams=2, image=128, max_cav=3, seed=100 + f)
        b["transformation_matrix"][0, 1] = b["transformation_matrix"][0, 1] @ torch.tensor(
            [[1, 0, 0, 1.5 * f], [0, 1, 0, -0.75 * f], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)   # pose differs per frame too
        out.append({k: v.to(cuda) for k, v in b.items()})
    return out


def _model(cuda):
    cfg = synth.corpbevt_small_config()
    return fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), cases.SEED).eval().to(cuda)


def test_captured_runner_equals_model_call(cuda):
    model = _model(cuda)
    frames = _frames(4, cuda)
    with host.compute_dtype(torch.bfloat16):
        ref = [{k: v.clone() for k, v in model(dict(f)).items()} for f in frames]
        run = pipeline.CapturedCorpBEVT(model, frames[0])
        assert run.graphs is not None
        for i in (1, 2, 3, 0, 2):
            out = run.step(frames[i])
            torch.cuda.synchronize()
            for k in ref[i]:
                assert torch.equal(out[k], ref[i][k]), "frame %d %s" % (i, k)
        # writing into the static buffers directly (what a data loader would do) is the same thing
        for k, dst in run.static_batch.items():              # (the keys CorpBEVT consumes; a batch may carry more)
            dst.copy_(frames[1][k] if k != "record_len" else frames[1][k].to(torch.int32))
        out = run.step()
        assert torch.equal(out["dynamic_seg"], ref[1]["dynamic_seg"])


@pytest.mark.parametrize("depth", [3, 4])
def test_pipelined_runner_equals_model_call(cuda, depth):
    model = _model(cuda)
    frames = _frames(9, cuda)
    with host.compute_dtype(torch.bfloat16):
        ref = [model(dict(f))["dynamic_seg"].clone() for f in frames]
        run = pipeline.PipelinedCorpBEVT(model, frames[0], depth=depth)
        assert run.latency_steps == depth
        got = []
        for i, f in enumerate(frames):
            out = run.step(f)
            got.append(None if out is None else out["dynamic_seg"].clone())
        for _ in range(depth - 1):                      # drain: resubmit the last frame
            got.append(run.step(frames[-1])["dynamic_seg"].clone())
        torch.cuda.synchronize()
    assert all(g is None for g in got[:depth - 1])
    for i in range(len(frames)):
        assert torch.equal(got[i + depth - 1], ref[i]), "frame %d came out wrong (depth %d)" % (i, depth)


def test_pipelined_runner_direct_writes_go_to_the_next_slot(cuda):
    """the documented loader pattern - write the next frame into `runner.static_batch[...]`, then `step()` - on the
    PIPELINED runner, whose camera matrices / poses / record_len live in a ring of slots: between steps `static_batch` must name
    the slot of the NEXT step, or the geometry of every frame lands one slot late"""
    model = _model(cuda)
    frames = _frames(7, cuda)
    depth = 3
    with host.compute_dtype(torch.bfloat16):
        ref = [model(dict(f))["dynamic_seg"].clone() for f in frames]
        run = pipeline.PipelinedCorpBEVT(model, frames[0], depth=depth)
        got = []
        for f in frames + [frames[-1]] * (depth - 1):
            for k, dst in run.static_batch.items():
                dst.copy_(f[k] if k != "record_len" else f[k].to(torch.int32))
            out = run.step()
            got.append(None if out is None else out["dynamic_seg"].clone())
        torch.cuda.synchronize()
    for i in range(len(frames)):
        assert torch.equal(got[i + depth - 1], ref[i]), "frame %d came out wrong" % i


def test_graph_plans_follow_weight_updates(cuda):
    """plans captured by `enable_graphs()` hold the addresses of the lowered weights; after an in-place parameter
    update (optimizer step / load_state_dict between validation passes) the plan is re-captured instead of replaying stale
    weights"""
    model = _model(cuda)
    f = _frames(1, cuda)[0]
    with host.compute_dtype(torch.bfloat16):
        model.enable_graphs()
        a = model(dict(f))["dynamic_seg"].clone()
        assert model.graph_plans.captures == 1
        model(dict(f))
        assert model.graph_plans.captures == 1
        with torch.no_grad():
            for p in model.seg_head.parameters():
                p.mul_(1.5)
        model.enable_graphs(False)
        want = model(dict(f))["dynamic_seg"].clone()
        model.enable_graphs()
        b = model(dict(f))["dynamic_seg"].clone()
        assert torch.equal(b, want) and not torch.equal(a, b)
        with torch.no_grad():
            for p in model.seg_head.parameters():
                p.mul_(1 / 1.5)
        assert model.graph_plans.captures == 1
        c = model(dict(f))["dynamic_seg"].clone()       # same plan cache, weights changed underneath it -> re-captured
        assert model.graph_plans.captures == 2
        model.enable_graphs(False)
        assert torch.equal(c, model(dict(f))["dynamic_seg"])
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.enable_graphs()
        model(dict(f))
        model.load_state_dict(sd)                        # invalidate_plans() empties the graph cache too
        assert len(model.graph_plans.plans) == 0


def test_captured_call_operator_level(cuda):
    """CapturedCall around SwapFusionEncoder.forward (the LiDAR bench workload's runner) == the eager call"""
    args = dict(input_dim=64, mlp_dim=128, agent_size=8, window_size=8, dim_head=32, drop_out=0.1, depth=2, mask=True)
    enc = fill_module_(host.SwapFusionEncoder(args), cases.SEED).eval().to(cuda)
    xs = [synth.procedural_input("cc.x", (1, 8, 64, 32, 32), s).to(cuda) for s in (0, 1)]
    mask = torch.ones(1, 32, 32, 1, 8, device=cuda)
    mask[0, :, :, :, 5:] = 0
    with host.compute_dtype(torch.bfloat16):
        ref = [enc(x, mask).clone() for x in xs]
        run = pipeline.CapturedCall(lambda a, m: enc(a, m), xs[0], mask)
        for i in (1, 0, 1):
            assert torch.equal(run.step(xs[i], mask), ref[i])


def test_agent_count_plans_serve_ragged_frames(cuda):
    """frames with 1, 2 and 3 agents arrive interleaved (record_len 1..max_cav, fuse_utils.py:8-61): one captured plan per agent
    count behind one step(), each result bit-identical to model(batch); the plan cache captures once per shape and evicts LRU"""
    model = _model(cuda)
    frames = {a: _frames(2, cuda, agents=a) for a in (1, 2, 3)}
    with host.compute_dtype(torch.bfloat16):
        ref = {(a, i): model(dict(f))["dynamic_seg"].clone() for a, fs in frames.items() for i, f in enumerate(fs)}
        srv = pipeline.AgentCountPlans(model, max_plans=2)
        order = [(2, 0), (1, 0), (2, 1), (3, 0), (1, 1), (3, 1), (2, 0)]
        for a, i in order:
            out = srv.step(frames[a][i])
            torch.cuda.synchronize()
            assert torch.equal(out["dynamic_seg"], ref[(a, i)]), "agents %d frame %d" % (a, i)
    # 3 shapes through a 2-plan LRU cache in this order: 2, 1 captured; 2 hit; 3 captured (evicts 1); 1 captured again (evicts 2);
    # 3 hit; 2 captured again (evicts 1)
    assert srv.captures == 5 and len(srv.plans) == 2


def test_pipelined_graph_holds_no_torch_copies(cuda):
    """the replayed step graph of the single-GPU pipeline is HIP kernels only - the per-frame camera matrices,
    poses and record_len live in host-filled ring slots, K/V land in their ring slots directly"""
    model = _model(cuda)
    frames = _frames(4, cuda)
    with host.compute_dtype(torch.bfloat16):
        run = pipeline.PipelinedCorpBEVT(model, frames[0], depth=3)
        for f in frames:
            run.step(f)
        torch.cuda.synchronize()
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                run.step()
            torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    kernels = [n for n in names if "Memcpy" not in n and "Memset" not in n]
    assert kernels, "profiler saw no device activity"
    bad = [n for n in names if "Memcpy" in n or "Memset" in n or "copyBuffer" in n or "at::native" in n or "fillBuffer" in n]
    assert not bad, "torch / runtime copies inside the replayed graph: %s" % bad


def test_model_call_served_from_graphs(cuda):
    """`model.enable_graphs()`: the plain eval-mode model(batch) call is served from a captured plan per frame shape and stays
    bit-identical to the eager forward; train() mode and a disabled switch go back to the eager paths"""
    model = _model(cuda)
    frames = {a: _frames(2, cuda, agents=a) for a in (1, 2)}
    with host.compute_dtype(torch.bfloat16):
        ref = {(a, i): model(dict(f))["dynamic_seg"].clone() for a, fs in frames.items() for i, f in enumerate(fs)}
        model.enable_graphs()
        for a, i in ((2, 0), (1, 1), (2, 1), (1, 0)):
            out = model(frames[a][i])
            torch.cuda.synchronize()
            assert torch.equal(out["dynamic_seg"], ref[(a, i)])
        assert model.graph_plans.captures == 2
        model.enable_graphs(False)
        assert torch.equal(model(dict(frames[2][0]))["dynamic_seg"], ref[(2, 0)])


# ----------------------------------------------------------------------------------------------------------------------
# uint8 ingest: camera frames as bytes, normalised inside the stem kernel (ResnetEncoder.set_rgb_normalisation,
# csrc/stem7x7.hip stem_pool_kernel<T, true>); reference: rgb_preprocessor.py:14-31 + inference_camera.py:56-61
# ----------------------------------------------------------------------------------------------------------------------
def _u8_frames(n, agents=2, bgr=False):
    out = []
    for f in range(n):
        b8, b32 = synth.opv2v_batch_u8(agents=agents, cams=2, image=128, max_cav=3, seed=300 + f, bgr=bgr)
        for b in (b8, b32):
            b["transformation_matrix"][0, 1] = b["transformation_matrix"][0, 1] @ torch.tensor(
                [[1, 0, 0, 1.25 * f], [0, 1, 0, -0.5 * f], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
        out.append((b8, b32))
    return out


@pytest.mark.parametrize("mode", [torch.float32, torch.bfloat16, "fp32_split", "fp32_fast"])
def test_uint8_ingest_equals_fp32_image_path_bit_for_bit(cuda, mode):
    """the model fed uint8 frames == the model fed the fp32 image the reference's pre-processor makes of those frames, bit for bit,
    in every compute mode (the stem looks the normalised value up in the pre-processor's own table); and the fp32-mode result is
    the oracle's on that fp32 image"""
    import oracle.corpbevt as o_model
    cfg = synth.corpbevt_small_config()
    model = fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), cases.SEED).eval()
    (b8, b32), = _u8_frames(1)
    ref = o_model.corpbevt_forward(model.state_dict(), cfg, b32)["dynamic_seg"] if mode == torch.float32 else None
    model = model.to(cuda)
    with host.compute_dtype(mode):
        with pytest.raises(CobevtHipError):            # bytes without a table: loud, not a silent cast
            model({k: v.to(cuda) for k, v in b8.items()})
        model.encoder.set_rgb_normalisation(synth.OPV2V_RGB_MEAN, synth.OPV2V_RGB_STD)
        a = model({k: v.to(cuda) for k, v in b32.items()})["dynamic_seg"].clone()
        b = model({k: v.to(cuda) for k, v in b8.items()})["dynamic_seg"].clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    if ref is not None:
        assert ((b.cpu() - ref).abs().max() / ref.abs().max()).item() <= 1e-3
    assert len(model.state_dict()) == len(host.CorpBEVT(copy.deepcopy(cfg)).state_dict())      # the table is not a checkpoint key


def test_uint8_ingest_bgr_frames(cuda):
    """bgr2rgb: true (corpbevt.yaml:28): BGR bytes in, the channel swap folded into the stem weights - equal to the fp32 path on the
    swapped, normalised image up to the summation order inside the stem (the three input channels sit at swapped K positions)"""
    model = _model(cuda)
    (b8, b32), = _u8_frames(1, bgr=True)
    with host.compute_dtype(torch.float32):
        a = model({k: v.to(cuda) for k, v in b32.items()})["dynamic_seg"].clone()
        model.encoder.set_rgb_normalisation(synth.OPV2V_RGB_MEAN, synth.OPV2V_RGB_STD, bgr2rgb=True)
        b = model({k: v.to(cuda) for k, v in b8.items()})["dynamic_seg"].clone()
    torch.cuda.synchronize()
    assert ((a - b).abs().max() / a.abs().max()).item() <= 1e-5
    assert not torch.equal(b, torch.zeros_like(b))


def test_host_frame_feeder_pulls_frames_inside_the_graph(cuda):
    """pinned uint8 frames through HostFrameFeeder (every step's graph pulls the NEXT frame's images from the pinned host ring with a
    fetch kernel, ops.host_fetch) give the frames' own outputs, in order, bit for bit - a different frame every step, so a pull
    from the wrong slot, a frame overwritten before it was pulled or consumed before it arrived shows up as a mismatch; frames are
    handed over both ways: copied into the ring by put(), and written in place into feeder.host_slot()"""
    model = _model(cuda)
    model.encoder.set_rgb_normalisation(synth.OPV2V_RGB_MEAN, synth.OPV2V_RGB_STD)
    frames = _u8_frames(8)
    depth = 3
    with host.compute_dtype(torch.bfloat16):
        ref = [model({k: v.to(cuda) for k, v in b8.items()})["dynamic_seg"].clone() for b8, _ in frames]
        pinned = [{k: v.pin_memory() for k, v in b8.items()} for b8, _ in frames]
        run = pipeline.PipelinedCorpBEVT(model, {k: v.to(cuda) for k, v in frames[0][0].items()}, depth=depth, host_ingest=True)
        assert run.slots[0]["inputs"].dtype == torch.uint8 and run.pinned[0].is_pinned()
        assert len({sl["inputs"].data_ptr() for sl in run.slots}) == depth
        for warm in range(4):                           # the feeder may start at any step of a running pipeline
            run.step()
        feeder = pipeline.HostFrameFeeder(run)
        got = []

        def hand_over(j):
            j = min(j, len(frames) - 1)                  # drain by resubmitting the last frame
            if j % 2:                                    # in place: the loader decodes straight into the ring slot
                slot = feeder.host_slot()
                slot.copy_(pinned[j]["inputs"])
                feeder.put(dict(pinned[j], inputs=slot))
            else:
                feeder.put(pinned[j])
        hand_over(0)
        for i in range(len(frames) + depth - 1):
            hand_over(i + 1)
            out = feeder.step()
            got.append(out["dynamic_seg"].clone())
        with pytest.raises(CobevtHipError):               # one frame ahead of the step in flight, not more
            feeder.put(pinned[0]); feeder.put(pinned[0])
        torch.cuda.synchronize()
    which = [[j for j in range(len(frames)) if torch.equal(got[i + depth - 1], ref[j])] for i in range(len(frames))]
    assert which == [[i] for i in range(len(frames))], "output i should be frame i's: %s" % which
    # the other order - put, step, put, step: the step's frame was not in the ring when the previous step pulled; still every
    # frame's own output (no overlap then), also when the two orders alternate
    with host.compute_dtype(torch.bfloat16):
        feeder = pipeline.HostFrameFeeder(run)
        got = []
        for i in range(len(frames) + depth - 1):
            j = min(i, len(frames) - 1)
            if not feeder.queue:
                feeder.put(pinned[j])
            if i % 3 == 0 and i + 1 < len(frames):        # ... and every third step also has its successor in the ring already
                feeder.put(pinned[i + 1])
            got.append(feeder.step()["dynamic_seg"].clone())
        torch.cuda.synchronize()
    which = [[j for j in range(len(frames)) if torch.equal(got[i + depth - 1], ref[j])] for i in range(len(frames))]
    assert which == [[i] for i in range(len(frames))], "late hand-over: output i should be frame i's: %s" % which
    with pytest.raises(CobevtHipError):
        pipeline.HostFrameFeeder(pipeline.PipelinedCorpBEVT(model, {k: v.to(cuda) for k, v in frames[0][0].items()}, depth=depth))


def test_host_fetch_kernel_copies_pinned_memory(cuda):
    from cobevt_amd import ops
    for n in (16, 4096 + 16, 3 * 1024 * 1024 + 48):
        src = torch.randint(0, 255, (n,), dtype=torch.uint8).pin_memory()
        dst = torch.zeros(n, dtype=torch.uint8, device=cuda)
        ops.host_fetch(src, dst)
        torch.cuda.synchronize()
        assert torch.equal(dst.cpu(), src)
    with pytest.raises(CobevtHipError):
        ops.host_fetch(torch.zeros(32, dtype=torch.uint8), torch.zeros(32, dtype=torch.uint8, device=cuda))    # pageable source
    # replayed from a captured graph while the host REWRITES the pinned buffer between replays (what the feeder does): every replay
    # must see the new bytes (system-scope loads: no line of the previous frame out of the GPU's caches)
    n = 2 * 1024 * 1024
    src = torch.zeros(n, dtype=torch.uint8).pin_memory()
    dst = torch.zeros(n, dtype=torch.uint8, device=cuda)
    run = pipeline.CapturedCall(lambda d: ops.host_fetch(src, d), dst)
    for it in range(6):
        fresh = torch.randint(0, 255, (n,), dtype=torch.uint8)
        src.copy_(fresh)
        out = run.step(None)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), fresh), "replay %d saw stale host bytes" % it
