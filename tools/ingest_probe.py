#!/usr/bin/env python
"""Where does the time of the H2D-inclusive loop go?  The three-frames-in-flight pipeline on uint8 frames, stepped
  (a) with the frames resident, (b) + the small per-frame tensors copied from pinned memory before every step, (c) + the image upload
  on the copy stream WITHOUT making the step wait for it, (d) the full copy-stream protocol (one step ahead), (e) the upload alone (no compute),
  (f) the upload through a raw hipMemcpyAsync on a non-blocking stream, (g) as (d) with the upload issued AFTER the replay;
  then variants of the protocol (which dependency costs what), and at the end the form the package ships: the pull INSIDE the captured
  graph (PipelinedCorpBEVT(host_ingest=True) + HostFrameFeeder).
Usage: python tools/ingest_probe.py [steps]"""
import copy
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobevt_amd import host, synth  # noqa: E402
from cobevt_amd.host import pipeline  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
host.set_compute_dtype(torch.bfloat16)
cfg = synth.corpbevt_config(max_cav=5)
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).eval().to(dev)
model.encoder.set_rgb_normalisation(synth.OPV2V_RGB_MEAN, synth.OPV2V_RGB_STD)
b8c, _ = synth.opv2v_batch_u8(agents=5, max_cav=5, seed=0)
b8 = {k: v.to(dev) for k, v in b8c.items()}
run = pipeline.PipelinedCorpBEVT(model, b8, depth=3, input_slots=True)
R = 4
pinned = []
for r in range(R):
    hb = {k: v.pin_memory() for k, v in b8c.items() if torch.is_tensor(v)}
    hb["record_len"] = b8c["record_len"].to(torch.int32).pin_memory()
    hb["inputs"] = torch.roll(b8c["inputs"], shifts=7 * r, dims=3).contiguous().pin_memory()
    pinned.append(hb)


def timed(step, warm=10):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


print("(a) resident                         %.4f ms/step" % timed(lambda: run.step()))
k = [0]


def small_only():
    hb = pinned[k[0] % R]
    q = run.i % run.depth
    small = {kk: hb[kk] for kk in run.slots[q] if kk != "inputs"}
    small["inputs"] = run.slots[q]["inputs"]
    run.step(small)
    k[0] += 1


print("(b) + small tensors from pinned host  %.4f ms/step" % timed(small_only))
copy_s = torch.cuda.Stream()


def upload_nowait():
    slot = (run.i + 1) % run.depth
    with torch.cuda.stream(copy_s):
        run.slots[slot]["inputs"].copy_(pinned[k[0] % R]["inputs"], non_blocking=True)
    run.step()
    k[0] += 1


print("(c) + image upload, step not waiting  %.4f ms/step" % timed(upload_nowait))
torch.cuda.synchronize()
_done, _up = [None] * 3, [None] * 3


def _upload1(j):
    slot = (run.i + (j - k[0])) % run.depth
    with torch.cuda.stream(copy_s):
        if _done[slot] is not None:
            copy_s.wait_event(_done[slot])
        run.slots[slot]["inputs"].copy_(pinned[j % R]["inputs"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(copy_s)
    _up[slot] = ev


_upload1(k[0])


def full():
    _upload1(k[0] + 1)
    q = run.i % run.depth
    hb = pinned[k[0] % R]
    sm = {kk: hb[kk] for kk in run.slots[q] if kk != "inputs"}
    sm["inputs"] = run.slots[q]["inputs"]
    torch.cuda.current_stream().wait_event(_up[q])
    run.step(sm)
    ev = torch.cuda.Event()
    ev.record()
    _done[q] = ev
    k[0] += 1


print("(d) copy stream, one step ahead       %.4f ms/step" % timed(full))
torch.cuda.synchronize()


def upload_only():
    with torch.cuda.stream(copy_s):
        run.slots[k[0] % 3]["inputs"].copy_(pinned[k[0] % R]["inputs"], non_blocking=True)
    k[0] += 1


print("(e) upload alone                      %.4f ms/step" % timed(upload_only))
hip = ctypes.CDLL("libamdhip64.so")
raw = ctypes.c_void_p()
assert hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1) == 0          # hipStreamNonBlocking
nbytes = b8c["inputs"].numel()


def raw_upload_nowait():
    slot = (run.i + 1) % run.depth
    rc = hip.hipMemcpyAsync(ctypes.c_void_p(run.slots[slot]["inputs"].data_ptr()), ctypes.c_void_p(pinned[k[0] % R]["inputs"].data_ptr()),
                            ctypes.c_size_t(nbytes), 1, raw)
    assert rc == 0
    run.step()
    k[0] += 1


print("(f) raw hipMemcpyAsync, not waiting   %.4f ms/step" % timed(raw_upload_nowait))
hip.hipStreamSynchronize(raw)


def upload_after():
    run.step()
    slot = run.i % run.depth                 # the NEXT step's slot (run.i already advanced)
    with torch.cuda.stream(copy_s):
        run.slots[slot]["inputs"].copy_(pinned[k[0] % R]["inputs"], non_blocking=True)
    k[0] += 1


print("(g) upload issued after the replay    %.4f ms/step" % timed(upload_after))


# ---- which part of the feeder protocol costs?  (variants that drop a dependency are for TIMING only: their outputs are racy)
class Variant(object):
    def __init__(self, main_waits=True, copy_waits=True, ahead=1, small=True, reuse_events=False):
        self.main_waits, self.copy_waits, self.ahead, self.small, self.reuse = main_waits, copy_waits, ahead, small, reuse_events
        D = run.depth
        self.up = [torch.cuda.Event() for _ in range(D)] if reuse_events else [None] * D
        self.done = [torch.cuda.Event() for _ in range(D)] if reuse_events else [None] * D
        self.live_up, self.live_done = [False] * D, [False] * D
        self.n = 0
        torch.cuda.synchronize()
        for a in range(ahead):
            self.upload(a)

    def upload(self, j):                       # frame index j -> slot of step run.i + (j - self.n)
        slot = (run.i + (j - self.n)) % run.depth
        with torch.cuda.stream(copy_s):
            if self.copy_waits and self.live_done[slot]:
                copy_s.wait_event(self.done[slot])
            run.slots[slot]["inputs"].copy_(pinned[j % R]["inputs"], non_blocking=True)
            ev = self.up[slot] if self.reuse else torch.cuda.Event()
            ev.record(copy_s)
        self.up[slot], self.live_up[slot] = ev, True

    def step(self):
        self.upload(self.n + self.ahead)
        q = run.i % run.depth
        hb = pinned[self.n % R]
        if self.main_waits and self.live_up[q]:
            torch.cuda.current_stream().wait_event(self.up[q])
        if self.small:
            sm = {kk: hb[kk] for kk in run.slots[q] if kk != "inputs"}
            sm["inputs"] = run.slots[q]["inputs"]
            run.step(sm)
        else:
            run.step()
        ev = self.done[q] if self.reuse else torch.cuda.Event()
        ev.record()
        self.done[q], self.live_done[q] = ev, True
        self.n += 1


for name, kw in (("(d) full protocol again", {}), ("(d1) main does not wait for the upload", dict(main_waits=False)),
                 ("(d2) copy does not wait for the consumer", dict(copy_waits=False)), ("(d3) no small tensors", dict(small=False)),
                 ("(d4) upload TWO steps ahead", dict(ahead=2)), ("(d5) events reused", dict(reuse_events=True)),
                 ("(d6) two ahead, no small tensors", dict(ahead=2, small=False))):
    torch.cuda.synchronize()
    v = Variant(**kw)
    print("%-42s %.4f ms/step" % (name, timed(v.step)))


# ---- does a per-step TIMING event (bench.py's timed_loop records one after every step for its median / p95) interact with the protocol?
def timed_with_events(step, warm=10):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(K):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


torch.cuda.synchronize()
v = Variant(ahead=2)
print("(d4) two ahead, plain loop                 %.4f ms/step" % timed(v.step))
print("(d4) two ahead, timing event per step      %.4f ms/step" % timed_with_events(v.step))
torch.cuda.synchronize()
print("resident, timing events                    %.4f ms/step" % timed_with_events(lambda: run.step()))

# ---- the shipped form: the pull inside the captured graph
run2 = pipeline.PipelinedCorpBEVT(model, b8, depth=3, host_ingest=True)
for r in range(3):
    run2.pinned[r].copy_(pinned[r]["inputs"])
f2 = pipeline.HostFrameFeeder(run2)
small = {kk: vv for kk, vv in pinned[0].items() if kk != "inputs"}
f2.put(dict(small, inputs=f2.host_slot()))


def fstep():
    f2.put(dict(small, inputs=f2.host_slot()))
    f2.step()


print("in-graph fetch kernel, plain loop          %.4f ms/step" % timed(fstep))
print("in-graph fetch kernel, timing events       %.4f ms/step" % timed_with_events(fstep))
# ... and with a dozen more streams alive in the process (what bench.py's process looks like)
extra = [torch.cuda.Stream() for _ in range(12)]
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
print("in-graph fetch kernel, 12 more streams     %.4f ms/step" % timed(fstep))
v = Variant(ahead=2)
print("copy stream two ahead, 12 more streams     %.4f ms/step" % timed(v.step))
