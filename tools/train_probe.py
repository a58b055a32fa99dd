"""Times the training slice on the headline shapes (GPU): the attention forward-with-lse and backward kernels alone, and one
train step (forward + backward + Adam) of the full-size swap-fusion encoders.  Prints one JSON object.

    python tools/train_probe.py            # on the GPU box
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import autograd as ag     # noqa: E402
from cobevt_amd import host, ops, synth   # noqa: E402


def _time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def attention_pair(name, mode, ncam, H, W, w, heads, batch, bias, masked):
    d = heads * 32
    tm = ops.tokmap(mode, ncam, H, W, w, w)
    rows = batch * ncam * H * W
    qkv = torch.randn(rows, 3 * d, device="cuda", requires_grad=True)
    table = torch.randn((2 * ncam - 1) * (2 * w - 1) ** 2, heads, device="cuda", requires_grad=True) if bias else None
    mask = torch.ones(batch, H, W, ncam, device="cuda") if masked else None
    wgt = torch.randn(rows, d, device="cuda")
    nq = ncam * w * w
    flops_f = 4.0 * batch * (H // w) * (W // w) * heads * nq * nq * 32

    def fwd():
        return ag.window_attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], tm, tm, tm, batch, heads, 0.17, rows,
                                   bias_table=table, bias_L=ncam, mask=mask)

    with torch.no_grad():
        t_f = _time(fwd)
    out = fwd()

    def bwd():
        qkv.grad = None
        out.backward(wgt, retain_graph=True)

    t_b = _time(bwd)
    # backward = 5 GEMM-shaped products (S, dP, dV, dK, dQ) vs 2 in the forward; includes the autograd glue (zeros, slice
    # gradient accumulation into d qkv)
    return {"case": name, "Nq": nq, "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4),
            "fwd_tflops": round(flops_f / t_f / 1e9, 2), "bwd_tflops": round(2.5 * flops_f / t_b / 1e9, 2)}


def encoder_step(name, args, shape, use_mask):
    enc = synth.fill_module_(host.SwapFusionEncoder(dict(args)), 0).train().cuda()
    b, l, d, h, w = shape
    x = torch.randn(shape, device="cuda")
    mask = torch.ones(b, h, w, 1, l, device="cuda") if use_mask else None
    target = torch.randn(b, d, h, w, device="cuda")
    opt = torch.optim.Adam(enc.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(enc(x, mask), target)
        loss.backward()
        opt.step()

    t = _time(step, iters=10, warm=2)
    enc.eval()
    with torch.no_grad(), host.compute_dtype(torch.float32):
        t_inf32 = _time(lambda: enc(x, mask), iters=10, warm=2)
    return {"case": name, "train_step_ms": round(t, 3), "fp32_inference_forward_ms": round(t_inf32, 3)}


def corpbevt_step(agents):
    """train_camera.py:143-179 on the full corpbevt.yaml model: forward, VanillaSegLoss, backward, Adam - fp32"""
    import copy
    cfg = synth.corpbevt_config()
    model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).train().cuda()
    batch = {k: v.cuda() for k, v in synth.opv2v_batch(agents=agents).items()}
    crit = host.VanillaSegLoss({"d_weights": 75.0, "s_weights": 15.0, "l_weights": 50, "d_coe": 2.0, "s_coe": 0.0, "target": "dynamic"})
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
    gt = None

    def step():
        nonlocal gt
        opt.zero_grad(set_to_none=True)
        out = model(dict(batch))
        if gt is None:
            gt = {"gt_dynamic": (torch.rand(out["dynamic_seg"].shape[:2] + out["dynamic_seg"].shape[3:], device="cuda") > 0.9).long(),
                  "gt_static": torch.zeros(out["dynamic_seg"].shape[:2] + out["dynamic_seg"].shape[3:], device="cuda", dtype=torch.long)}
        loss = crit(out, gt)
        loss.backward()
        opt.step()
        return float(loss.detach())

    torch.cuda.reset_peak_memory_stats()
    l0 = step()
    t = _time(step, iters=5, warm=1)
    l1 = step()
    res = {"case": "CorpBEVT corpbevt.yaml, %d agents x 4 cams x 512x512, fp32 train step (forward + VanillaSegLoss + backward + AdamW)" % agents,
           "train_step_ms": round(t, 2), "loss_first": round(l0, 4), "loss_after_8_steps": round(l1, 4),
           "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    # the same step inside a bf16 autocast region (train_camera.py --half with bf16): convolutions on the bf16 kernels, library
    # GEMMs in bf16, the attention / LayerNorm / GELU / loss kernels in fp32 behind casts
    def step_amp():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(dict(batch)), gt)
        loss.backward()
        opt.step()
        return float(loss.detach())
    la = step_amp()
    ta = _time(step_amp, iters=5, warm=1)
    res.update({"bf16_autocast_train_step_ms": round(ta, 2), "bf16_autocast_loss_first": round(la, 4), "bf16_autocast_loss_after_7_steps": round(step_amp(), 4)})
    return res


def main():
    out = {"attention": [], "encoder": [], "model": []}
    if os.environ.get("PROBE_ONLY") == "model":            # same-job A/B runs: only the whole-model steps
        for agents in (2, 5):
            out["model"].append(corpbevt_step(agents))
        print(json.dumps(out, indent=1))
        return
    out["attention"].append(attention_pair("fusion window 5 agents 32x32 w8", 0, 5, 32, 32, 8, 4, 1, True, True))
    out["attention"].append(attention_pair("fusion grid 5 agents 32x32 w8", 1, 5, 32, 32, 8, 4, 1, True, True))
    out["attention"].append(attention_pair("LiDAR window 8 agents 256x256 w8", 0, 8, 256, 256, 8, 2, 1, True, True))
    cam = synth.corpbevt_config()["fax_fusion"] if "fax_fusion" in synth.corpbevt_config() else None
    if cam is not None:
        cam = dict(cam, drop_out=0.0)
        out["encoder"].append(encoder_step("camera FuseBEVT (5 x 128 x 32 x 32, depth %d)" % cam["depth"], cam,
                                           (1, cam["agent_size"], cam["input_dim"], 32, 32), cam.get("mask", False)))
    lidar = dict(input_dim=64, mlp_dim=128, agent_size=8, window_size=8, dim_head=32, drop_out=0.0, depth=3, mask=True)
    out["encoder"].append(encoder_step("LiDAR FuseBEVT (8 x 64 x 256 x 256, depth 3, BASELINE configs[4])", lidar,
                                       (1, 8, 64, 256, 256), True))
    for agents in (2, 5):
        out["model"].append(corpbevt_step(agents))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
