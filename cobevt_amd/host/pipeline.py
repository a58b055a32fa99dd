"""HIP-graph runners for CorpBEVT inference: what `bench.py` times and what a serving loop uses.

The reference's inference loop (opv2v/opencood/tools/inference_camera.py:38-60) calls `model(batch_data['ego'])` once per
frame on the default stream.  The eager `CorpBEVT.forward` mirror does the same through ~105 ctypes launches; these runners
keep that forward bit for bit and remove the launch overhead / expose the frame-level parallelism:

* `CapturedCorpBEVT(model, example_batch)`     one frame at a time, the whole forward replayed from captured HIP graphs
  (encode | [agent all-gather between the two graphs when world > 1] | fuse + decode).  `step(batch)` == `model(batch)`.
* `PipelinedCorpBEVT(model, example_batch, depth=3)`   several frames in flight on ONE GPU: step i runs the camera encoder
  + K/V sides of frame i, the FAX query path of frame i-1 and fusion + decoder of frame i-2 on three HIP streams inside one
  captured graph.  One frame in, one frame out per step; `step(batch)` returns the output of the frame submitted
  `latency_steps - 1` calls earlier (None while the pipeline fills).  Nothing is skipped or cached.

Both take the batch through STATIC device buffers (`runner.static_batch`): a captured graph reads fixed addresses, so
`step(batch)` first copies the caller's tensors into them (skipped for tensors that already ARE the static buffers - a
data loader can write into `runner.static_batch[...]` directly).  Multi-GPU (`rank`, `world`): see cobevt_amd/dist.py.
"""
import os

import torch

from .. import dist as cdist
from .. import ops
from ..lib import CobevtHipError
from .runtime import GraphOwner

_IMAGE_KEYS = ("inputs", "intrinsic", "extrinsic")


def _static_copy(batch, dev):
    out = {k: batch[k].to(dev).clone() for k in _IMAGE_KEYS}
    out["transformation_matrix"] = batch["transformation_matrix"].to(device=dev, dtype=torch.float32).clone()
    out["record_len"] = torch.as_tensor(batch["record_len"]).to(device=dev, dtype=torch.int32).clone()
    return out


class _RunnerBase(GraphOwner):
    def __init__(self, model, example_batch, rank=0, world=1, agents=None):
        if model.training:
            raise CobevtHipError("graph runners implement inference: call model.eval() first")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise CobevtHipError("graph runners need the model on a ROCm device")
        self.model, self.rank, self.world = model, rank, world
        self.static_batch = _static_copy(example_batch, dev)
        self.agents = int(self.static_batch["inputs"].shape[0]) if agents is None else int(agents)

    @property
    def _images(self):
        return {k: self.static_batch[k] for k in _IMAGE_KEYS}

    def load(self, batch):
        """copy the caller's frame into the static input buffers (no-op for tensors that already are those buffers)"""
        if batch is None:
            return
        for k, dst in self.static_batch.items():
            src = batch[k]
            src = torch.as_tensor(src) if not torch.is_tensor(src) else src
            if src.data_ptr() == dst.data_ptr():
                continue
            if tuple(src.shape) != tuple(dst.shape):
                raise CobevtHipError("graph runner captured %s of shape %s, got %s (re-capture for a new shape)"
                                     % (k, tuple(dst.shape), tuple(src.shape)))
            if k == "inputs" and src.dtype != dst.dtype:     # uint8 camera frames vs the normalised fp32 image: different stem kernels
                raise CobevtHipError("graph runner captured %s images, got %s (re-capture for the other input format)" % (dst.dtype, src.dtype))
            dst.copy_(src, non_blocking=True)


class CapturedCorpBEVT(_RunnerBase):
    """One frame at a time from captured HIP graphs; `step(batch)` equals `model(batch)` bit for bit."""

    latency_steps = 1

    def __init__(self, model, example_batch, rank=0, world=1, agents=None, use_graph=True):
        super().__init__(model, example_batch, rank, world, agents)
        self.graphs, self.out = None, None
        self.eager_step()                       # builds every weight plan outside the capture
        torch.cuda.synchronize()
        if use_graph:
            self.capture()

    def _encode(self):
        return self.model.encode_agents(dict(self._images))

    def _fuse(self, feats):
        return self.model.fuse_and_decode(feats, self.static_batch["transformation_matrix"], self.static_batch["record_len"])

    def eager_step(self, batch=None):
        self.load(batch)
        feats = self._encode()
        mine = cdist.exchange_features(feats, self.rank, self.world, self.agents)
        self.out = self._fuse(mine)
        return self.out

    def capture(self):
        """one HIP graph for the whole frame on one GPU; with world > 1 encode and fuse are two graphs and the collective stays eager
        between them"""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        if self.world == 1:                   # nothing between the two halves: ONE graph, one replay (a replay boundary is ~20 us)
            with torch.cuda.graph(g1):
                self.feats = self.fuse_in = self._encode()
                self.out = self._fuse(self.fuse_in)
            self.graphs = (g1,)
            return
        with torch.cuda.graph(g1):
            feats = self._encode()
        self.feats = feats
        self.fuse_in = torch.empty_like(feats)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            self.out = self._fuse(self.fuse_in)
        self.graphs = (g1, g2)

    def step(self, batch=None):
        if self.graphs is None:
            return self.eager_step(batch)
        self.load(batch)
        self.graphs[0].replay()
        if self.world > 1:
            cdist.exchange_features(self.feats, self.rank, self.world, self.agents, out=self.fuse_in)
            self.graphs[1].replay()
        return self.out


class AgentCountPlans(object):
    """One captured plan per frame shape behind ONE `step(batch)`.  Real OPV2V frames carry 1..max_cav agents
    (intermediate_fusion_dataset.py:261-295 concatenates the agents present, `record_len` says how many; regroup pads to
    max_cav, fuse_utils.py:8-61), while a captured HIP graph fixes (agents, cameras, image size).  A serving loop therefore
    keeps one `CapturedCorpBEVT` per shape it has seen - captured on first use from that frame, replayed afterwards - and
    dispatches on the incoming batch.  `step(batch)` equals `model(batch)` bit for bit for every agent count.

    max_plans bounds the cache (least recently used plan dropped; its graphs and static buffers are freed).

    A captured graph holds the raw addresses of the lowered weight buffers (`HipModule._plan`), and those buffers are rebuilt
    when a parameter changes.  Every plan therefore remembers the (version counter, address) of every parameter and
    floating-point buffer of the model at capture time; `step` compares them and re-captures the plan after an optimizer
    step, `load_state_dict` or any other in-place update (train_camera.py's train -> eval validation loop).  Writes through
    `.data` bypass the version counter: call `model.invalidate_plans()` after those, which also empties this cache."""

    def __init__(self, model, use_graph=True, max_plans=8, prewarm=None):
        self.model, self.use_graph, self.max_plans = model, use_graph, int(max_plans)
        self.plans = {}            # key -> (weight fingerprint, CapturedCorpBEVT), insertion order = recency
        self._epoch = -1
        self._tensors = None       # the model's parameters / floating-point buffers, listed once (578 tensors over 483 modules on the
        #                            full CorpBEVT: walking the module tree every step cost ~2 ms of host time per frame)
        self.captures = 0
        self.busy = False          # True while a plan is being built (the model's own forward runs eagerly inside)
        for b in (prewarm or []):  # capture ahead of time (e.g. one synthetic frame per agent count) instead of on first sight
            self._runner(b)

    @staticmethod
    def key(batch):
        rl = batch["record_len"]
        n_scen = int(rl.numel()) if torch.is_tensor(rl) else len(rl)
        from . import runtime as rt
        return (tuple(batch["inputs"].shape), str(batch["inputs"].dtype), n_scen, tuple(batch["transformation_matrix"].shape),
                rt.get_compute_mode())

    def weights_fingerprint(self):
        """(version counter, address) of every parameter / floating-point buffer the captured kernels may read"""
        from . import runtime as rt
        if self._tensors is None or self._epoch != rt.structure_epoch():
            self._tensors, self._epoch = rt.module_tensors(self.model), rt.structure_epoch()
        return tuple((t._version, t.data_ptr()) for t in self._tensors)

    def clear(self):
        """drop every plan and the cached tensor list (HipModule.invalidate_plans calls this: after weight surgery through `.data`,
        or when parameters / buffers were added, removed or replaced)"""
        self.plans.clear()
        self._tensors = None

    def _runner(self, batch):
        k = self.key(batch)
        fp = self.weights_fingerprint()
        ent = self.plans.pop(k, None)
        if ent is not None and ent[0] != fp:       # weights moved on since the capture: the graph reads dead buffers
            ent = None
        if ent is None:
            self.busy = True
            try:
                r = CapturedCorpBEVT(self.model, batch, use_graph=self.use_graph)
            finally:
                self.busy = False
            self.captures += 1
            while len(self.plans) >= self.max_plans:
                self.plans.pop(next(iter(self.plans)))
            ent = (fp, r)
        self.plans[k] = ent
        return ent[1]

    def step(self, batch):
        return self._runner(batch).step(batch)


class PipelinedCorpBEVT(_RunnerBase):
    """Several frames in flight on ONE GPU.  A CoBEVT frame is ~1.2 ms of camera encoder that fills the chip followed by
    ~1 ms of FAX query path, swap fusion and decoder whose ~100 dependent launches are latency-bound and leave most CUs idle.
    Step i therefore runs, on separate HIP streams inside one captured graph,
        S1  = encoder + K/V sides of the FAX pyramid of frame i          (CorpBEVT.encode_trunk)
        S2  = FAX query path of frame i-1                                 (CorpBEVT.fax_query)         [depth 3]
              or S2a = pyramid level 0 of frame i-1 and S2b = levels 1.. + global attention of frame i-2 [depth 4]
        S3  = STTF + swap fusion + decoder + head of the oldest frame     (CorpBEVT.fuse_and_decode)
    with the state that crosses steps (projected K/V of the three levels, the level-0 output, the (A,32,32,128) features,
    the poses) in rings of `depth` buffers - `depth` graphs, replayed round-robin.  Every step takes one frame in and
    completes one frame; the latency of a frame is `latency_steps` steps (depth on one GPU; one more with the agent
    all-gather of cobevt_amd/dist.py, which runs under the following step)."""

    def __init__(self, model, example_batch, rank=0, world=1, agents=None, depth=3, input_slots=False, host_ingest=False):
        """input_slots: one image buffer per ring slot instead of one shared buffer.
        host_ingest (implies input_slots): every step also PULLS THE NEXT FRAME'S IMAGES out of a ring of pinned host buffers
        (`self.pinned[slot]`, one per ring slot) with a fetch kernel captured in the step's graph (ops.host_fetch): step k reads
        pinned slot (k + 1) % depth into image slot (k + 1) % depth while its own kernels run.  `HostFrameFeeder` is the
        host-side protocol around it."""
        super().__init__(model, example_batch, rank, world, agents)
        if depth not in (3, 4):
            raise CobevtHipError("PipelinedCorpBEVT: depth must be 3 or 4")
        self.depth = D = depth
        self.host_ingest = bool(host_ingest)
        self.input_slots = bool(input_slots) or self.host_ingest
        # The small per-frame inputs (camera matrices, poses, record_len) are consumed up to `depth` steps after the images, so
        # they live in RINGS of `depth` slots that `load()` fills from the host side: graph q reads slot q for its encoder
        # stage and the slots of the earlier frames for its later stages - no copy of them inside the replayed graph.
        # `static_batch` (the public "write your next frame here" dict) always points at the slot of the NEXT step.
        sb = self.static_batch
        self.slots = [sb] + [{k: v.clone() for k, v in sb.items() if k != "inputs"} for _ in range(D - 1)]
        for sl in self.slots[1:]:                             # the images are consumed within the step: one buffer (or one per slot)
            sl["inputs"] = sb["inputs"].clone() if self.input_slots else sb["inputs"]
        self.pinned = None
        if self.host_ingest:
            src = sb["inputs"].cpu()
            self.pinned = [src.clone().pin_memory() for _ in range(D)]
        st = model.encode_trunk(self._images_of(0))
        torch.cuda.synchronize()
        self.meta = [{k: v for k, v in lvl.items() if not torch.is_tensor(v)} for lvl in st["kv"]]
        self.batch = st["batch"]
        self.kv = [[{k: torch.empty_like(v) for k, v in lvl.items() if torch.is_tensor(v)} for lvl in st["kv"]] for _ in range(D)]
        # E_inv of slot q as encode_trunk derives it: a view of the ring slot when the extrinsics are fp32 (OPV2V passes them
        # un-inverted, fax_modules.py:502-503), so the later stages of later steps read the slot itself
        self.einv = [model.fax._extrinsic(sl["extrinsic"].reshape(-1, 4, 4).to(torch.float32)).contiguous() for sl in self.slots]
        x0 = model.fax_query(st, levels=(0, 1))
        self.x = [torch.empty_like(x0) for _ in range(D)] if depth == 4 else None
        feats = model.fax_query(st, levels=(1, len(st["kv"])), x=x0)
        self.f = [torch.empty_like(feats) for _ in range(D)]
        # multi-GPU: the frame's agents are gathered (one RCCL all-gather between graph replays) into g; single GPU: g is f.
        # The gather of step q's features runs on its own stream UNDER step q+1 and is consumed by the fusion stage of step
        # q+2 (one more step of latency than on one GPU), so the collective is off the critical path
        self.g = self.f if world == 1 else [torch.empty_like(feats) for _ in range(D)]
        self.lag = 1 if world == 1 else 2                     # fusion of step q reads the features of slot q - lag
        self.latency_steps = D if world == 1 else D + 1
        # with the gather's extra step the pose slot fusion needs is the one `load()` rewrites for the same step -> the graph
        # parks it in a staged copy first (multi-GPU only)
        self.pose_s3 = [sl["transformation_matrix"].clone() for sl in self.slots] if world > 1 else None
        self.rlen_s3 = [sl["record_len"].clone() for sl in self.slots] if world > 1 else None
        self.comm = torch.cuda.Stream() if world > 1 else None
        self.gathered = [None] * D
        self.out, self.graphs = None, None
        self.i = self.filled = 0
        self.capture()

    def _images_of(self, slot):
        return {k: self.slots[slot][k] for k in _IMAGE_KEYS}

    @property
    def _next_slot(self):
        return self.i % self.depth

    def load(self, batch):
        """copy the caller's frame into the input buffers of the next step's slot (no-op for tensors that already are them)"""
        self.static_batch = self.slots[self._next_slot]
        super().load(batch)

    def _state(self, slot):
        return {"kv": [dict(self.meta[i], **self.kv[slot][i]) for i in range(len(self.meta))],
                "E_inv": self.einv[slot], "batch": self.batch}

    def _s1(self, slot):
        # A/B knob (tools only): COBEVT_PIPE_GATE="s3@1,s2@2" makes the encoder's stage AFTER ResNet stage k (0..3) wait for the named
        # branch of this step - do the small-launch stages collide less with the layer-3 / 4 convolutions, which own their CUs,
        # when they are confined to the first half of the step?  (profiles/r06_pipe_gate_ab.txt)
        gate = os.environ.get("COBEVT_PIPE_GATE")
        hook = None
        if gate and self.depth == 3:
            names = {"s2": self.streams[0], "s3": self.streams[1]}
            plan = [(names[t.split("@")[0]], int(t.split("@")[1])) for t in gate.split(",")]

            def hook(stage):
                for strm, k in plan:
                    if k == stage:
                        torch.cuda.current_stream().wait_stream(strm)
        st = self.model.encode_trunk(self._images_of(slot), kv_out=self.kv[slot], stage_hook=hook)      # K/V land in the slot directly
        main = torch.cuda.current_stream()
        for level, lvl in enumerate(st["kv"]):
            main.wait_stream(st["side"][level])
            for k, v in lvl.items():
                if torch.is_tensor(v) and v.data_ptr() != self.kv[slot][level][k].data_ptr():
                    self.kv[slot][level][k].copy_(v)
        e = st["E_inv"]
        if e.data_ptr() != self.slots[slot]["extrinsic"].data_ptr():     # encode_trunk had to convert it: keep the result per slot
            if self.einv[slot] is None or self.einv[slot].data_ptr() == self.slots[slot]["extrinsic"].data_ptr():
                self.einv[slot] = torch.empty_like(e)
            self.einv[slot].copy_(e)
        else:
            self.einv[slot] = e                                           # a view of the ring slot: nothing to copy

    def _s3(self, q):
        """fusion + decoder of the frame whose images were submitted latency_steps - 1 steps before step q"""
        feats = self.g[(q - self.lag) % self.depth]
        if self.world == 1:
            src = self.slots[(q - (self.latency_steps - 1)) % self.depth]       # filled by load() latency_steps - 1 steps ago
            return self.model.fuse_and_decode(feats, src["transformation_matrix"], src["record_len"])
        return self.model.fuse_and_decode(feats, self.pose_s3[q], self.rlen_s3[q])

    def _exchange(self, q):
        """after step q: gather f[q] into g[q] on the communication stream (it waits for the step, the next step does not
        wait for it)"""
        if self.world > 1:
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                cdist.exchange_features(self.f[q], self.rank, self.world, self.agents, out=self.g[q])
                ev = torch.cuda.Event()
                ev.record(self.comm)
            self.gathered[q] = ev

    def _await_gather(self, q):
        """before step q: its fusion stage reads the features gathered after step q - lag"""
        if self.world > 1:
            ev = self.gathered[(q - self.lag) % self.depth]
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def _step_body(self, q):
        """slot q = step index mod depth: S1 writes kv[q]; the later stages read the slots written 1, 2, .. steps ago"""
        D = self.depth
        main = torch.cuda.current_stream()
        if self.world > 1:        # (see __init__) the frame fused in step q + 1 was loaded into slot q + 1 `depth` steps ago
            nxt = (q + 1) % D
            self.pose_s3[nxt].copy_(self.slots[nxt]["transformation_matrix"])
            self.rlen_s3[nxt].copy_(self.slots[nxt]["record_len"])
        for s in self.streams:
            s.wait_stream(main)
        nlev = len(self.meta)

        def pull():
            # the next step's images: pinned host ring -> image slot, beside this step's kernels.  Issued on stage 3's stream, IN
            # FRONT of it: that branch (fusion + decoder, ~0.3 ms of small launches) has a millisecond of slack in the step, whereas
            # a branch of its own was serialised in front of the encoder by the graph executor (+0.33 ms per step = the pull's own
            # duration).  Same-job A/B of placement and size (gpurun_out/r05j, resident 1.73 ms per step on that box): in front of
            # stage 3 with 32 workgroups 1.89 ms, behind it 1.95 (32) / 2.00 (128) / 2.03 (512), behind stage 2 2.02
            if self.host_ingest:
                nxt = (q + 1) % D
                ops.host_fetch(self.pinned[nxt], self.slots[nxt]["inputs"], blocks=16)     # 16: profiles/r06_ingest_split_ab.txt (0.918 vs 0.888 with 32)

        # A/B knob (tools only; unset = the placement above): COBEVT_INGEST_PLAN="s3f:0.5,s2f:0.5" pulls byte shares of the frame at
        # several points of the step - s3f / s3b = in front of / behind stage 3 on its stream, s2f / s2b the same for stage 2, s1f in
        # front of the encoder; COBEVT_INGEST_BLOCKS = workgroups per pull (profiles/r06_ingest_split_ab.txt)
        plan_env = os.environ.get("COBEVT_INGEST_PLAN") if self.host_ingest else None
        shares = {}
        if plan_env:
            nxt = (q + 1) % D
            src8, dst8 = self.pinned[nxt].view(-1).view(torch.uint8), self.slots[nxt]["inputs"].view(-1).view(torch.uint8)
            total, lo = src8.numel(), 0
            items = [(t.split(":")[0], float(t.split(":")[1])) for t in plan_env.split(",")]
            for j, (pos, frac) in enumerate(items):
                hi = total if j == len(items) - 1 else min(total, (lo + int(total * frac)) // 4096 * 4096)
                shares.setdefault(pos, []).append((lo, hi))
                lo = hi
            nblk = int(os.environ.get("COBEVT_INGEST_BLOCKS", "16"))

            def pull_at(pos):
                for a, b in shares.get(pos, []):
                    if b > a:
                        ops.host_fetch(src8[a:b], dst8[a:b], blocks=nblk)
        else:
            def pull_at(pos):
                if pos == "s3f":
                    pull()
        # Capture order = dispatch order of the graph's roots, and it matters: oldest frame first (stage 3, stage 2, then the encoder)
        # 640-650 frames/s on the box of profiles/r05_ab_same_job.txt, stage 2 before stage 3 the same (641-657), the encoder first
        # 579-586 (its full-chip workgroups then starve the later stages' small dependent launches, whose chains end up as the step's tail)
        if D == 3:
            s2, s3 = self.streams
            with torch.cuda.stream(s3):
                pull_at("s3f")
                out = self._s3(q)
                pull_at("s3b")
            with torch.cuda.stream(s2):                                       # frame i-1 -> its features into slot q
                pull_at("s2f")
                self.model.fax_query(self._state((q - 1) % D), joined=False, out=self.f[q])
                pull_at("s2b")
        else:
            s2a, s2b, s3 = self.streams
            with torch.cuda.stream(s3):
                pull_at("s3f")
                out = self._s3(q)
            with torch.cuda.stream(s2b):                                      # frame i-2: K/V from two steps ago, x from one
                self.model.fax_query(self._state((q - 2) % D), joined=False, levels=(1, nlev), x=self.x[(q - 1) % D],
                                     out=self.f[q])
            with torch.cuda.stream(s2a):                                      # frame i-1
                self.model.fax_query(self._state((q - 1) % D), joined=False, levels=(0, 1), out=self.x[q])
        pull_at("s1f")
        self._s1(q)
        for s in self.streams:
            main.wait_stream(s)
        return out

    def capture(self):
        D = self.depth
        # (high-priority streams for the later stages - so that their small launches are dispatched ahead of the encoder's full-chip
        #  workgroups - measured +0.1-0.2 % in two same-job pairs, i.e. nothing: profiles/r05_ab_same_job.txt)
        self.streams = tuple(torch.cuda.Stream() for _ in range(D - 1))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for it in range(2 * D):
                self._await_gather(it % D)
                self._step_body(it % D)
                self._exchange(it % D)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graphs, self.outs = [], []
        pool = None
        for q in range(D):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):           # (capturing on a high-priority stream - the encoder's - measured nothing either)
                self.outs.append(self._step_body(q))
            pool = g.pool()
            self.graphs.append(g)
        self.i = self.filled = 0
        self.static_batch = self.slots[0]

    def step(self, batch=None):
        """submit one frame, complete one frame.  Returns the output dict of the frame submitted latency_steps - 1 calls ago
        (buffers the same graph overwrites `depth` steps later), or None while the pipeline is still filling."""
        self.load(batch)
        q = self.i % self.depth
        self._await_gather(q)
        self.graphs[q].replay()
        self._exchange(q)
        self.out = self.outs[q]
        self.i += 1
        self.filled += 1
        self.static_batch = self.slots[self._next_slot]      # the public dict names the NEXT step's slot between steps
        return self.out if self.filled >= self.latency_steps else None


class HostFrameFeeder(object):
    """Camera frames from PINNED host memory into a `PipelinedCorpBEVT(..., host_ingest=True)`: the ingest of the reference's loop
    (inference_camera.py:56-61 moves every frame's batch to the device before the forward) as part of the captured step.

        feeder = HostFrameFeeder(pipe)
        feeder.put(frame_0)
        for k in ...:
            feeder.put(frame_k+1)           # into the pinned ring (or decode straight into feeder.host_slot(): no copy at all)
            out = feeder.step()             # step k computes on frame k and pulls frame k+1 over PCIe beside its kernels

    put() the next frame BEFORE step() as above and the transfer is hidden; put() after step() (put, step, put, step) is still
    correct - the step notices that its frame was not in the ring when the previous step pulled, and copies it in stream order - but
    then nothing overlaps.

    How the bytes travel: a fetch kernel inside the step's HIP graph (csrc/elementwise.hip host_fetch_kernel, 32 workgroups of
    system-scope 8-byte loads from the device-visible pinned buffer) - no copy engine, no extra stream, no event between replays.
    The stream-based form (hipMemcpyAsync on a copy stream + events) was built first and measured (tools/ingest_probe.py,
    profiles/r05_ingest_probe.txt): ROCm maps a process's streams onto 4 hardware queues, and whenever the copy
    stream shared one with a branch of the step's graph the transfer sat IN FRONT of that branch's kernels - step + copy time
    (2.22 ms instead of 1.64) in bench.py's process, free in a process with fewer streams, worse with a high-priority stream or
    more queues; a replay waiting for a one-step-old event was held back another ~0.2 ms.  Inside the graph there is nothing to
    collide with.  Only the images come from the ring: camera matrices / poses / record_len (a few hundred bytes, still read by
    later pipeline stages of running steps) are copied in stream order right before their step.  put() copies THEM too - into a
    feeder-owned ring of pinned host sets, one per frame handed over (queued by reference, a loader that reuses its
    buffers for the next frame overwrote frame k's poses before step(k) uploaded them, and could race the pending asynchronous
    copy after it) - so the caller may reuse every buffer of `host_batch` as soon as put() returns.  A set is rewritten only after
    the event behind the step that uploaded it.  uint8 frames (`ResnetEncoder.set_rgb_normalisation`) make a 5-agent frame 15.7 MB instead of 63."""

    def __init__(self, runner):
        if not isinstance(runner, PipelinedCorpBEVT) or not runner.host_ingest:
            raise CobevtHipError("HostFrameFeeder needs a PipelinedCorpBEVT built with host_ingest=True")
        self.r = runner
        self.base = runner.i                          # step index of frame 0
        self.fetched = [None] * runner.depth          # event: every device read of this pinned slot issued so far has run
        self.queue = []                               # small tensors of the frames handed over and not yet stepped
        self.n_put = 0
        self.pulled = False                           # did the previous step's graph pull the frame the next step() computes on?
        self.small_sets = [None] * (2 * runner.depth)  # feeder-owned pinned copies of the small tensors, one set per frame handed over
        self.small_used = [None] * (2 * runner.depth)  # event behind the step that uploaded the set

    def _stage_small(self, host_batch):
        """the frame's small tensors copied into the next pinned set of the ring (non-tensors pass through)"""
        i = self.n_put % len(self.small_sets)
        if self.small_used[i] is not None:
            self.small_used[i].synchronize()          # the step that uploaded this set has read it
            self.small_used[i] = None
        cur = self.small_sets[i] or {}
        out = {}
        for k in self.r.slots[0]:
            if k == "inputs":
                continue
            v = host_batch[k]
            if torch.is_tensor(v):
                buf = cur.get(k)
                if not torch.is_tensor(buf) or buf.shape != v.shape or buf.dtype != v.dtype:
                    buf = torch.empty(v.shape, dtype=v.dtype, pin_memory=torch.cuda.is_available())
                buf.copy_(v)
                v = buf
            out[k] = v
        self.small_sets[i] = out
        return i, out

    def host_slot(self):
        """the pinned buffer the NEXT put() frame belongs in - a loader may decode straight into it and pass it to put().
        Waits until the device has finished reading the frame that lived there (`depth` frames ago)."""
        slot = (self.base + self.n_put) % self.r.depth
        if self.fetched[slot] is not None:
            self.fetched[slot].synchronize()
        return self.r.pinned[slot]

    def put(self, host_batch):
        r = self.r
        if len(self.queue) >= 2:
            raise CobevtHipError("HostFrameFeeder: one frame ahead of the step in flight (put, put, step, put, step, ...)")
        src = host_batch["inputs"]
        ref = r.pinned[0]
        if tuple(src.shape) != tuple(ref.shape) or src.dtype != ref.dtype:
            raise CobevtHipError("HostFrameFeeder.put: captured %s %s, got %s %s" % (tuple(ref.shape), ref.dtype, tuple(src.shape), src.dtype))
        dst = self.host_slot()
        if src.data_ptr() != dst.data_ptr():
            dst.copy_(src)                            # host memcpy into the ring (skipped when the loader wrote in place)
        if self.n_put == 0:
            # nothing pulls the very first frame: fetch it now, in stream order, while the pipeline is not stepping yet (the same
            # transfer issued between two replays of a running loop is the late path of step(); profiles/r05_ab_same_job.txt)
            slot = self.base % r.depth
            ops.host_fetch(dst, r.slots[slot]["inputs"])
            self.fetched[slot] = torch.cuda.Event()
            self.fetched[slot].record()               # the put() that rewrites this ring slot (`depth` frames on) waits for this read
            self.pulled = True
        self.queue.append(self._stage_small(host_batch))
        self.n_put += 1

    def step(self):
        r = self.r
        if not self.queue:
            raise CobevtHipError("HostFrameFeeder.step: put() a frame first")
        small_set, small = self.queue.pop(0)
        small = dict(small)
        q = r.i % r.depth
        late = not self.pulled
        if late:      # the first frame, or one put() after the previous step was launched: that step's pull read the slot too early.
            #           The same fetch kernel in stream order, not a copy-engine transfer (see the class docstring)
            ops.host_fetch(r.pinned[q], r.slots[q]["inputs"])
        small["inputs"] = r.slots[q]["inputs"]                      # in place already: load() skips it
        self.pulled = bool(self.queue)                              # the next frame is in the ring while this step pulls it
        out = r.step(small)
        ev = torch.cuda.Event()
        ev.record()
        self.fetched[(q + 1) % r.depth] = ev                        # this step pulled pinned slot q + 1
        self.small_used[small_set] = ev                             # ... and uploaded this set of small tensors
        if late:
            self.fetched[q] = ev
        return out


class CapturedCall(GraphOwner):
    """`fn(*args)` replayed from one captured HIP graph; the arguments are held in static device buffers (`self.args`,
    refreshed by `step(*new_args)`), the result tensors are the graph's own output buffers.  Used for the operator-level
    workloads (LiDAR FuseBEVT `SwapFusionEncoder`, nuScenes SinBEVT) whose forward is a single stage."""

    def __init__(self, fn, *args, use_graph=True, warmup=2):
        self.fn = fn
        self.args = tuple(a.clone() if torch.is_tensor(a) else a for a in args)
        self.out = fn(*self.args)               # builds the weight plans outside the capture
        torch.cuda.synchronize()
        self.graph = None
        if use_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    fn(*self.args)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = fn(*self.args)

    def step(self, *args):
        for dst, src in zip(self.args, args):
            if torch.is_tensor(dst) and src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.graph is None:
            self.out = self.fn(*self.args)
        else:
            self.graph.replay()
        return self.out


class FrameShardedCorpBEVT(GraphOwner):
    """Strong scaling ("latency mode"; BASELINE.json north_star: one agent per GPU, ONE all-gather before
    FuseBEVT): ONE frame over `world` GPUs.  Rank r encodes the agents r, r + world, .. of the frame (its sub-batch is
    `static_sub`), one exchange hands every rank all agents' (H, W, C) features, and the 18-GF STTF + fusion + decoder tail
    runs replicated on every rank (cheaper than a second exchange).

    gather = "rccl":   torch.distributed.all_gather_into_tensor (RCCL over xGMI), eager between two captured graphs;
    gather = "direct": the one-shot peer-window exchange (cobevt_amd.dist.DirectExchange, csrc/peer_gather.hip) - two more
                       kernels INSIDE the captured graph, so a step is a single graph replay.
    depth = 1: encode -> exchange -> fuse, the latency of a frame is one step.
    depth = 2: step i runs encode + exchange of frame i and, on a second stream of the same graph, fusion + decoder of frame
               i - 1 (two alternating windows / buffers); one frame in, one frame out per step, latency two steps."""

    #: direct gather: EVERY step enqueues an asynchronous read of the windows' status words behind the step (pinned host
    #: buffer + event, no synchronisation) and looks at the reads that have completed: a bounded flag wait that gave up raises
    #: one or two steps later - a stalled or missing rank must not turn into a run of normal-looking outputs fused from a
    #: partially filled window.  The raise is local to this rank: the caller must tear the runner down collectively (the ranks'
    #: epochs are out of step).  `status_every` > 0 adds the synchronising check (`status()`) after the first step and every so
    #: many steps; `status()` at the end of a run is the caller's (bench.py does).
    status_every = 0

    def __init__(self, model, sub_batch, frame_batch, rank, world, agents, use_graph=True, gather="rccl", depth=1, group=None):
        if model.training:
            raise CobevtHipError("graph runners implement inference: call model.eval() first")
        if gather not in ("rccl", "direct") or depth not in (1, 2):
            raise CobevtHipError("FrameShardedCorpBEVT: gather must be 'rccl' or 'direct', depth 1 or 2")
        dev = next(model.parameters()).device
        self.model, self.rank, self.world, self.agents = model, rank, world, int(agents)
        self.gather, self.depth, self.group = gather, depth, group
        self.latency_steps = depth
        self.mine = cdist.agents_of_rank(rank, world, self.agents)
        self.static_sub = {k: sub_batch[k].to(dev).clone() for k in _IMAGE_KEYS}
        pose = frame_batch["transformation_matrix"].to(device=dev, dtype=torch.float32)
        rlen = torch.as_tensor(frame_batch["record_len"]).to(device=dev, dtype=torch.int32)
        self.pose = [pose.clone() for _ in range(depth)]            # slot q % depth holds the pose of the frame submitted at step q
        self.rlen = [rlen.clone() for _ in range(depth)]
        self.graphs, self.out, self.outs = None, None, [None] * depth
        feats = self._encode()                  # also for a surplus rank (one agent): fixes the block shape, builds the plans
        block = tuple(feats.shape[1:])
        self.slots = cdist.slots_per_rank(world, self.agents)
        if gather == "direct":
            self.windows = [cdist.DirectExchange(block, feats.dtype, self.agents, rank, world, group=group, device=dev)
                            .plan(*cdist.direct_plan_strong(rank, world, self.agents)) for _ in range(depth)]
            self.full = [w.window for w in self.windows]
            self.staging = None
        else:
            self.windows = None
            self.staging = torch.zeros((self.slots,) + block, device=dev, dtype=feats.dtype)
            self.full = [torch.zeros((self.agents,) + block, device=dev, dtype=feats.dtype) for _ in range(depth)]
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.barrier(group=group)      # plans are built: ranks enter the first exchange together
        self.empty = feats[:0]
        self.feats = [self.empty] * depth        # slot q: the features the encoder of step q produced (graph q's own buffer)
        self.i = self.filled = 0
        self.side = torch.cuda.Stream(device=dev) if depth == 2 else None
        for q in range(2 * depth):
            self._step_body(q % depth, eager=True)
        torch.cuda.synchronize()
        if use_graph:
            self.capture()

    def _encode(self):
        return self.model.encode_agents(dict(self.static_sub))

    def _exchange(self, feats, slot):
        if self.gather == "direct":
            self.windows[slot](feats)
        else:
            cdist.exchange_features_strong(feats, len(self.mine), self.rank, self.world, self.agents, group=self.group,
                                           out=self.full[slot], staging=self.staging)

    def _fuse(self, slot):
        return self.model.fuse_and_decode(self.full[slot], self.pose[slot], self.rlen[slot])

    def _step_body(self, q, eager=False, stage=None):
        """slot q = step index mod depth.  stage None: the whole step; "a": everything in front of an eager (RCCL) exchange;
        "b": everything behind it (depth 1 only: the fusion of the same frame)."""
        main = torch.cuda.current_stream()
        if self.depth == 1:
            if stage in (None, "a"):
                self.feats[q] = self._encode() if self.mine else self.empty
            if stage is None:
                self._exchange(self.feats[0], 0)
            if stage in (None, "b"):
                self.outs[0] = self._fuse(0)
            return
        # depth 2: fusion of the previous frame (slot q - 1) on the side stream, under this frame's encoder
        prev = (q - 1) % 2
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self.outs[prev] = self._fuse(prev)
        self.feats[q] = self._encode() if self.mine else self.empty
        if stage is None:
            self._exchange(self.feats[q], q)
        main.wait_stream(self.side)

    def eager_step(self):
        q = self.i % self.depth
        self._step_body(q, eager=True)
        return self._finish(q)

    def capture(self):
        in_graph = self.gather == "direct"            # the peer-window kernels are captured; an RCCL call stays eager
        self.graphs = []
        pool = None
        for q in range(self.depth):
            stages = [None] if in_graph else (["a", "b"] if self.depth == 1 else ["a"])
            gs = []
            for st in stages:
                if st == "a" and self.depth == 1 and not self.mine:
                    gs.append(None)                  # a surplus rank has nothing in front of the exchange
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    self._step_body(q, stage=st)
                pool = g.pool()
                gs.append(g)
            self.graphs.append(gs)

    def _finish(self, q):
        self.i += 1
        self.filled += 1
        if self.windows:
            for w in self.windows:
                st = w.status_poll()
                if st:
                    raise CobevtHipError("peer-window exchange timed out (status %d): a rank is missing or stalled; rebuild the runner "
                                         "on every rank" % st)
                w.status_async()
            if self.status_every and (self.i == 1 or self.i % self.status_every == 0):
                self.status()
        self.out = self.outs[(q - (self.depth - 1)) % self.depth]
        return self.out if self.filled >= self.latency_steps else None

    def load(self, batch, q):
        """copy the caller's frame (this rank's agents' images, the frame's poses) into the static buffers of slot q"""
        if batch is None:
            return
        for k in _IMAGE_KEYS:
            if batch[k].data_ptr() != self.static_sub[k].data_ptr():
                self.static_sub[k].copy_(batch[k], non_blocking=True)
        self.pose[q].copy_(batch["transformation_matrix"], non_blocking=True)
        self.rlen[q].copy_(torch.as_tensor(batch["record_len"]), non_blocking=True)

    def step(self, batch=None):
        q = self.i % self.depth
        self.load(batch, q)
        if self.graphs is None:
            self._step_body(q, eager=True)
            return self._finish(q)
        gs = self.graphs[q]
        if self.gather == "direct":
            gs[0].replay()
        else:
            if gs[0] is not None:
                gs[0].replay()
            self._exchange(self.feats[q], q)
            if self.depth == 1:
                gs[1].replay()
        return self._finish(q)

    def status(self):
        """direct gather only: raise if a bounded flag wait gave up (a peer never arrived).  After a timeout the ranks' epochs
        are out of step: the runner must be rebuilt (collectively) before it is used again."""
        if self.windows:
            for w in self.windows:
                st, _ = w.status()
                if st:
                    raise CobevtHipError("peer-window exchange timed out (status %d): a rank is missing or stalled" % st)
