#!/usr/bin/env python
"""Times cobevt_amd.host.train_graph.CapturedTrainStep (one optimisation step of train_camera.py:143-179 captured into a HIP graph)
against the eager step, fp32 and inside a bf16 autocast region.

History: round 2 measured 47.0 -> 43.9 ms (2 agents, fp32) and kept the class out of the package because replays differed from eager
steps by up to 1.5 % of the update norm at one stride-2 encoder block.  Round 3 traced that kind of difference to isolated ReLU flips
(tools/train_grad_diag.py, tools/bn_flip_probe.py) - two eager runs with 1e-7 perturbations show it too - so the class now
lives in the package with a test that bounds the drift (tests/test_training_gpu.py::test_captured_train_step_follows_eager).

Usage on the GPU box:  python tools/train_graph_probe.py [agents]
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import host, synth  # noqa: E402
from cobevt_amd.host.train_graph import CapturedTrainStep  # noqa: E402


def _time(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


if __name__ == "__main__":
    agents = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = synth.corpbevt_config()
    model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).train().cuda()
    batch = {k: v.cuda() for k, v in synth.opv2v_batch(agents=agents).items()}
    crit = host.VanillaSegLoss({"d_weights": 75.0, "s_weights": 15.0, "l_weights": 50, "d_coe": 2.0, "s_coe": 0.0, "target": "dynamic"})
    with torch.no_grad():
        shp = model.eval()(dict(batch))["dynamic_seg"].shape
    model.train()
    batch["gt_dynamic"] = (torch.rand(shp[:2] + shp[3:], device="cuda") > 0.9).long()
    batch["gt_static"] = torch.zeros(shp[:2] + shp[3:], device="cuda", dtype=torch.long)
    del model

    out = {}
    for name, dt in (("fp32", None), ("bf16_autocast", torch.bfloat16)):
        model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).train().cuda()
        opt = torch.optim.AdamW(model.parameters(), lr=2e-4, capturable=True)

        def eager():
            opt.zero_grad(set_to_none=True)
            with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt is not None):
                loss = crit(model(dict(batch)), batch)
            loss.backward()
            opt.step()
        t_eager = _time(eager, 5, 2)
        cap = CapturedTrainStep(model, lambda o, b: crit(o, b), opt, batch, autocast_dtype=dt)
        t_graph = _time(lambda: cap.step(), 10, 2)
        out[name] = {"eager_ms": round(t_eager, 2), "captured_ms": round(t_graph, 2), "loss": round(float(cap.step()), 4)}
        del cap, opt, model
    print("CorpBEVT corpbevt.yaml, %d agents, AdamW step: %s" % (agents, out))
