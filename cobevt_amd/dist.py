"""Agent-sharded multi-GPU execution of CoBEVT: one process per GPU, one all-gather of per-agent BEV features
(RCCL over xGMI; `backend="nccl"` is RCCL on ROCm) before FuseBEVT — the V2V feature-sharing step.

The reference runs every agent in one process (intermediate_fusion_dataset.py:289-295 batches the agents;
corpbevt.py:112-124 treats them as a batch dimension until `regroup`), so this exchange is new code.
Sharding scheme ("weak scaling"): with W ranks, W frames are in flight per step; the W*A agent tasks
t = f*A + a are dealt round-robin, task t -> rank t % W (slot t // W), so every rank encodes A agents that belong to
up to A different frames; ONE all-gather of the (A, H, W, C) feature blocks follows, rank f picks frame f's A
agents out of the gathered tensor and runs STTF + swap fusion + decoder for its frame.  At W = 1 this is the plain
single-GPU forward.  The bookkeeping below is pure integer arithmetic, testable on CPU with gloo.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun convention).
    Returns (rank, world, local_rank).  World size 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # COBEVT_DIST_BACKEND=gloo: dry-run the multi-rank control flow with several ranks sharing one GPU (a one-GPU box
        # cannot host two RCCL ranks); the measured configuration is always RCCL
        backend = device_backend or os.environ.get("COBEVT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (required by the host driver for RCCL)
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)                        # bind the GPU before the communicator is created
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def tasks_of_rank(rank, world, agents):
    """[(frame, agent)] encoded by `rank`, in slot order (W frames x `agents` agents dealt round-robin)."""
    return [divmod(slot * world + rank, agents) for slot in range(agents)]


def gather_index(frame, world, agents):
    """flat indices into the all-gathered (world * agents) feature blocks holding agents 0..A-1 of `frame`."""
    idx = []
    for a in range(agents):
        t = frame * agents + a
        idx.append((t % world) * agents + (t // world))
    return idx


_index_cache = {}


def _gather_index_tensor(rank, world, agents, device):
    """gather_index(...) as a device tensor, uploaded once: a per-step host->device copy from pageable memory would make
    the host wait for the stream (i.e. for the previous graph replay) before it can enqueue the next step."""
    key = (rank, world, agents, str(device))
    idx = _index_cache.get(key)
    if idx is None:
        idx = torch.tensor(gather_index(rank, world, agents), dtype=torch.long).to(device)
        _index_cache[key] = idx
    return idx


def exchange_features(local_feats, rank, world, agents, group=None, out=None):
    """local_feats: (agents, ...) features of this rank's tasks -> (agents, ...) features of frame `rank` (written into
    `out` when given).  One all-gather; works for CPU (gloo) and device (RCCL) tensors."""
    if world == 1:
        if out is not None and out.data_ptr() != local_feats.data_ptr():
            out.copy_(local_feats)
            return out
        return local_feats
    assert local_feats.shape[0] == agents
    local_feats = local_feats.contiguous()
    gathered = torch.empty((world * agents,) + tuple(local_feats.shape[1:]), dtype=local_feats.dtype,
                           device=local_feats.device)
    if local_feats.is_cuda and dist.get_backend(group) == "gloo":     # dry-run mode only: gloo gathers through the host
        host = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(host, local_feats.cpu(), group=group)
        gathered.copy_(host)
    else:
        dist.all_gather_into_tensor(gathered, local_feats, group=group)
    idx = _gather_index_tensor(rank, world, agents, local_feats.device)
    if out is not None:
        return torch.index_select(gathered, 0, idx, out=out)
    return gathered.index_select(0, idx)


class AgentShardedCoBEVT(object):
    """Step function of the sharded pipeline around any object offering encode_agents(batch) and
    fuse_and_decode(feats, transformation_matrix, record_len) (cobevt_amd.host.CorpBEVT; tests inject the oracle)."""

    def __init__(self, model, rank, world, agents, group=None):
        self.model, self.rank, self.world, self.agents, self.group = model, rank, world, agents, group

    def step(self, task_batch, frame_pose, record_len):
        """task_batch: inputs / intrinsic / extrinsic of THIS rank's `agents` tasks; frame_pose: (1, max_cav, 4, 4)
        transformation matrices of frame `rank`; returns the output dict of frame `rank`."""
        feats = self.model.encode_agents(task_batch)
        mine = exchange_features(feats, self.rank, self.world, self.agents, self.group)
        return self.model.fuse_and_decode(mine, frame_pose, record_len)


# ----------------------------------------------------------------------------------------------
# Strong scaling ("latency mode"): ONE frame, its agents spread over the ranks (rank r owns the agents
# {a : a mod G = r}; one all-gather of the (H, W, C) feature blocks; STTF + fusion + decoder replicated on every rank -
# 18 GF, cheaper than a second exchange).  With more ranks than agents the surplus ranks contribute zero blocks.
# ----------------------------------------------------------------------------------------------
def agents_of_rank(rank, world, agents):
    """agent ids encoded by `rank` for one frame"""
    return list(range(rank, agents, world))


def slots_per_rank(world, agents):
    """feature blocks every rank contributes to the all-gather (equal sizes; ranks with fewer agents pad with zeros)"""
    return max(1, -(-agents // world))


def strong_gather_index(world, agents):
    """flat indices into the all-gathered (world * slots) blocks holding agents 0..A-1 of the frame"""
    s = slots_per_rank(world, agents)
    return [(a % world) * s + a // world for a in range(agents)]


def take_agents(batch, ids):
    """sub-batch (inputs / intrinsic / extrinsic) of the agents `ids` of a one-frame batch"""
    idx = torch.as_tensor(ids, dtype=torch.long, device=batch["inputs"].device)
    return {k: batch[k].index_select(0, idx) for k in ("inputs", "intrinsic", "extrinsic")}


_strong_index_cache = {}


def exchange_features_strong(local_feats, n_local, rank, world, agents, group=None, out=None, staging=None):
    """local_feats: (>= n_local, H, W, C) features of this rank's agents (agents_of_rank order) -> (agents, H, W, C) features
    of the whole frame in agent order on EVERY rank.  One all-gather of `slots_per_rank` blocks per rank."""
    if world == 1:
        if out is not None and out.data_ptr() != local_feats.data_ptr():
            out.copy_(local_feats)
            return out
        return local_feats
    s = slots_per_rank(world, agents)
    block = tuple(local_feats.shape[1:])
    send = staging if staging is not None else torch.zeros((s,) + block, dtype=local_feats.dtype, device=local_feats.device)
    if n_local:
        send[:n_local].copy_(local_feats[:n_local])
    if n_local < s:
        send[n_local:].zero_()
    gathered = torch.empty((world * s,) + block, dtype=send.dtype, device=send.device)
    if send.is_cuda and dist.get_backend(group) == "gloo":            # dry-run mode only
        host = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        gathered.copy_(host)
    else:
        dist.all_gather_into_tensor(gathered, send, group=group)
    key = (world, agents, str(send.device))
    idx = _strong_index_cache.get(key)
    if idx is None:
        idx = torch.tensor(strong_gather_index(world, agents), dtype=torch.long).to(send.device)
        _strong_index_cache[key] = idx
    if out is not None:
        return torch.index_select(gathered, 0, idx, out=out)
    return gathered.index_select(0, idx)


# ----------------------------------------------------------------------------------------------
# One-shot direct exchange over xGMI: 256-KiB agent blocks are latency-bound in a ring, so every rank
# stores its blocks straight into the peers' windows (csrc/peer_gather.hip, include/cobevt_hip.h cobevt_peer_*).
# ----------------------------------------------------------------------------------------------
class _DeviceBytes(object):
    """raw device memory as a __cuda_array_interface__ provider (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class DirectExchange(object):
    """Window-based replacement of the agent all-gather for DEVICE tensors.

    `block_shape` / `dtype`: one agent's feature block; `blocks`: block slots of the window (the frame's agents in agent
    order).  Set-up (collective over `group`, any backend): allocate the window, all-gather the hipIpc handles, map the
    peers.  `plan(dest_rank, dest_block)`: where this rank's local blocks go (dest_rank < 0: every rank).  `__call__(local)`
    enqueues one exchange on the current stream and returns the window as an (blocks, *block_shape) tensor - valid for the
    kernels that follow in stream order, until the next exchange is enqueued.  Capturable in a HIP graph."""

    def __init__(self, block_shape, dtype, blocks, rank, world, group=None, device=None, spin_limit=0):
        import ctypes
        from . import lib as _L
        self._L, self._ct = _L, ctypes
        self.rank, self.world, self.blocks = int(rank), int(world), int(blocks)
        self.block_shape, self.dtype = tuple(block_shape), dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        n = 1
        for d in self.block_shape:
            n *= int(d)
        self.block_bytes = n * torch.empty((), dtype=dtype).element_size()
        if self.block_bytes % 16:
            raise ValueError("DirectExchange: a block must be a multiple of 16 bytes")
        self.window_bytes = self.block_bytes * self.blocks
        self.spin_limit = int(spin_limit)
        lib = _L.load()
        ptr, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            _L.check(lib.cobevt_peer_window_alloc(self.window_bytes, ctypes.byref(ptr), handle), "cobevt_peer_window_alloc")
        self._own = ptr.value
        handles = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(handles, (os.getpid(), handle.raw), group=group)
        else:
            handles[0] = (os.getpid(), handle.raw)
        self._mapped = []
        ptrs = (ctypes.c_void_p * self.world)()
        for r, (pid, raw) in enumerate(handles):
            if r == self.rank:
                ptrs[r] = self._own
                continue
            q = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                _L.check(lib.cobevt_peer_window_open(ctypes.create_string_buffer(raw, 64), ctypes.byref(q)),
                         "cobevt_peer_window_open (rank %d's window)" % r)
            ptrs[r] = q.value
            self._mapped.append(q.value)
        self._ptrs = ptrs
        raw_t = torch.as_tensor(_DeviceBytes(self._own, self.window_bytes), device=self.device)
        self.window = raw_t.view(dtype).reshape((self.blocks,) + self.block_shape)
        self._dest_rank = self._dest_block = None
        self._n_local = 0
        if self.world > 1:
            dist.barrier(group=group)            # every peer has mapped every window before the first store

    def plan(self, dest_rank, dest_block):
        if len(dest_rank) != len(dest_block) or len(dest_rank) > 16:
            raise ValueError("DirectExchange.plan: at most 16 local blocks")
        self._n_local = len(dest_rank)
        self._dest_rank = (self._ct.c_int * max(1, self._n_local))(*[int(v) for v in dest_rank])
        self._dest_block = (self._ct.c_int * max(1, self._n_local))(*[int(v) for v in dest_block])
        return self

    def __call__(self, local):
        """local: (>= n_local, *block_shape) contiguous device tensor (ignored when this rank sends nothing)"""
        n = self._n_local
        if n:
            if not local.is_cuda or not local.is_contiguous() or local.dtype != self.dtype or local.shape[0] < n:
                raise ValueError("DirectExchange: local blocks must be a contiguous device tensor of the window's dtype")
        lib = self._L.load()
        stream = self._ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._L.check(lib.cobevt_peer_exchange(self._ct.c_void_p(local.data_ptr()) if n else None, self._ptrs, self.world,
                                               self.rank, n, self.block_bytes, self._dest_rank, self._dest_block,
                                               self.window_bytes, self.spin_limit, stream), "cobevt_peer_exchange")
        return self.window

    def status(self):
        """(status, completed exchanges) after synchronising the current stream; status != 0: a bounded wait gave up"""
        st, ep = self._ct.c_int(), self._ct.c_int()
        stream = self._ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._L.check(self._L.load().cobevt_peer_window_status(self._ct.c_void_p(self._own), self.window_bytes,
                                                               self._ct.byref(st), self._ct.byref(ep), stream),
                      "cobevt_peer_window_status")
        return st.value, ep.value

    _st_host = None

    def status_async(self):
        """enqueue a read of the window's status words behind everything issued so far on the current stream (no synchronisation);
        `status_poll()` looks at the reads that have completed"""
        if self._st_host is None:
            self._st_host = [torch.zeros(32, dtype=torch.int32).pin_memory() for _ in range(4)]
            self._st_ev, self._st_n = [None] * 4, 0
        j = self._st_n % 4
        if self._st_ev[j] is not None and not self._st_ev[j].query():
            return                                     # four reads already in flight: the oldest will report
        stream = self._ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._L.check(self._L.load().cobevt_peer_window_status_async(self._ct.c_void_p(self._own), self.window_bytes,
                                                                     self._ct.c_void_p(self._st_host[j].data_ptr()), stream),
                      "cobevt_peer_window_status_async")
        ev = torch.cuda.Event()
        ev.record()
        self._st_ev[j] = ev
        self._st_n += 1

    def status_poll(self):
        """largest status word among the asynchronous reads that have completed (0: every bounded wait they saw had completed; the
        word is sticky, so a timeout is reported by every later read)"""
        worst = 0
        if self._st_host is not None:
            for j in range(4):
                if self._st_ev[j] is not None and self._st_ev[j].query():
                    worst = max(worst, int(self._st_host[j][18]))
        return worst

    def close(self):
        lib = self._L.load()
        torch.cuda.synchronize(self.device)
        for q in self._mapped:
            lib.cobevt_peer_window_close(self._ct.c_void_p(q))
        self._mapped = []
        if self._own:
            self.window = None
            lib.cobevt_peer_window_free(self._ct.c_void_p(self._own))
            self._own = None


def direct_plan_strong(rank, world, agents):
    """latency mode: this rank's agents r, r + G, .. go to every rank, block slot = agent id"""
    mine = agents_of_rank(rank, world, agents)
    return [-1] * len(mine), list(mine)


def direct_plan_weak(rank, world, agents):
    """throughput mode: local slot j holds task t = j * world + rank = (frame t // agents, agent t % agents) and goes to the
    frame's owner only (an all-to-all, not an all-gather: each block crosses one link once)"""
    tasks = tasks_of_rank(rank, world, agents)
    return [f for f, _ in tasks], [a for _, a in tasks]


class FrameShardedCoBEVT(object):
    """One frame across the ranks (strong scaling): rank r encodes agents r, r+G, ..; all-gather; fusion replicated."""

    def __init__(self, model, rank, world, agents, group=None):
        self.model, self.rank, self.world, self.agents, self.group = model, rank, world, agents, group
        self.mine = agents_of_rank(rank, world, agents)
        self._zero_block = None

    def step(self, frame_batch):
        """frame_batch: the WHOLE frame's batch dict (every rank sees the same host-side inputs; only its agents' images
        are read).  Returns the frame's output dict on every rank."""
        feats = None
        if self.mine:
            feats = self.model.encode_agents(take_agents(frame_batch, self.mine))
        if feats is None:             # a surplus rank (more GPUs than agents): contributes zero blocks of the right shape
            if self._zero_block is None:      # learn shape / dtype / device once by encoding one agent
                ref = self.model.encode_agents(take_agents(frame_batch, [0]))
                self._zero_block = ref.new_zeros((0,) + tuple(ref.shape[1:]))
            feats = self._zero_block
        full = exchange_features_strong(feats, len(self.mine), self.rank, self.world, self.agents, self.group) \
            if self.world > 1 else feats
        return self.model.fuse_and_decode(full, frame_batch["transformation_matrix"], frame_batch["record_len"])


# ----------------------------------------------------------------------------------------------
# LiDAR FuseBEVT (SwapFusionEncoder on (1, L, C, H, W) with L agents, SURVEY.md §8e "larger fusion inputs"): the window
# pass is independent per window and the dilated-grid pass per grid group, so the map is sharded by ROWS:
#   band layout  rank r owns rows [r*H/G, (r+1)*H/G)                       - whole windows (H/G a multiple of w)
#   grid layout  rank s owns rows {i*X + x : x in [s*X/G, (s+1)*X/G)}, X = H/w - whole grid groups; as a local map of
#                w*X/G rows in (i, x_local) order this is again a plain dilated grid of stride X/G
# with one all-to-all between the two halves of every block (and one back), an agent->band all-to-all in front (rank a
# holds agent a's map: "agent-per-GPU") and an all-gather of the fused bands at the end.
# ----------------------------------------------------------------------------------------------
def _all_to_all(send, group=None):
    recv = torch.empty_like(send)
    if send.is_cuda and dist.get_backend(group) == "gloo":            # dry-run mode only
        h = torch.empty(send.shape, dtype=send.dtype)
        dist.all_to_all_single(h, send.cpu(), group=group)
        recv.copy_(h)
    else:
        dist.all_to_all_single(recv, send, group=group)
    return recv


def shard_check(H, window, world, agents):
    X = H // window
    if H % window or window % world or X % world or agents % world:
        raise ValueError("row sharding needs window %% world == 0, (H / window) %% world == 0 and agents %% world == 0 "
                         "(H=%d window=%d agents=%d world=%d)" % (H, window, agents, world))


def agents_to_bands(x_agents, world, group=None):
    """(b, L/G, H, W, d) maps of this rank's agents [r*L/G, (r+1)*L/G) -> (b, L, H/G, W, d) band r of every agent"""
    if world == 1:
        return x_agents
    b, ll, H, W, d = x_agents.shape
    send = x_agents.reshape(b, ll, world, H // world, W, d).permute(2, 0, 1, 3, 4, 5).contiguous()
    recv = _all_to_all(send, group)                                   # (src, b, ll, Hb, W, d)
    return recv.permute(1, 0, 2, 3, 4, 5).reshape(b, world * ll, H // world, W, d)


def bands_to_grid(x_band, world, window, group=None):
    """band layout (b, L, H/G, W, d) -> grid layout (b, L, w*X/G, W, d), rows in (i, x_local) order"""
    if world == 1:
        return x_band
    b, l, Hb, W, d = x_band.shape
    XG = Hb // window                           # X / G rows per (i, rank) block: X/G = (H/w)/G = Hb/w
    wi = window // world                        # i values per band
    send = x_band.reshape(b, l, wi, world, XG, W, d).permute(3, 0, 1, 2, 4, 5, 6).contiguous()
    recv = _all_to_all(send, group)                                   # (src, b, l, wi, X/G, W, d)
    return recv.permute(1, 2, 0, 3, 4, 5, 6).reshape(b, l, window * XG, W, d)


def grid_to_bands(x_grid, world, window, group=None):
    """inverse of bands_to_grid"""
    if world == 1:
        return x_grid
    b, l, Hg, W, d = x_grid.shape
    XG = Hg // window
    wi = window // world
    send = x_grid.reshape(b, l, world, wi, XG, W, d).permute(2, 0, 1, 3, 4, 5, 6).contiguous()
    recv = _all_to_all(send, group)                                   # (src s, b, l, wi, X/G, W, d)
    return recv.permute(1, 2, 3, 0, 4, 5, 6).reshape(b, l, wi * world * XG, W, d)


def mask_band(mask, rank, world):
    """(b, H, W, 1, L) -> this rank's band rows"""
    Hb = mask.shape[1] // world
    return mask[:, rank * Hb:(rank + 1) * Hb].contiguous()


def mask_grid(mask, rank, world, window):
    """(b, H, W, 1, L) -> this rank's grid-layout rows ((i, x_local) order)"""
    b, H, W, e, l = mask.shape
    X = H // window
    XG = X // world
    return mask.reshape(b, window, X, W, e, l)[:, :, rank * XG:(rank + 1) * XG].reshape(b, window * XG, W, e, l).contiguous()


class RowShardedFuseBEVT(object):
    """SwapFusionEncoder over row-sharded maps.  `stages`: list of (mode, fn) in network order with
    fn(x_local (b, L, h_local, W, d), mask_local (b, h_local, W, 1, L) | None) -> x_local, mode 0 = window pass (band
    layout), 1 = grid pass (grid layout); `head(x_band) -> (b, Hb, W, d)`.  The HIP product passes its own stage functions
    (cobevt_amd.host.swap_fusion_modules.sharded_stages); the CPU tests inject the oracle."""

    def __init__(self, stages, head, rank, world, window, group=None):
        self.stages, self.head, self.rank, self.world, self.window, self.group = stages, head, rank, world, window, group

    def step(self, x_agents, mask):
        """x_agents (b, L/G, H, W, d): this rank's agents; mask (b, H, W, 1, L) replicated or None -> (b, H, W, d) fused"""
        G, w = self.world, self.window
        if G > 1:
            shard_check(x_agents.shape[2], w, G, x_agents.shape[1] * G)
        x = agents_to_bands(x_agents, G, self.group)
        mb = mask_band(mask, self.rank, G) if mask is not None else None
        mg = mask_grid(mask, self.rank, G, w) if mask is not None else None
        layout = 0
        for mode, fn in self.stages:
            if mode != layout:
                x = bands_to_grid(x, G, w, self.group) if mode == 1 else grid_to_bands(x, G, w, self.group)
                layout = mode
            x = fn(x, mb if mode == 0 else mg)
        if layout == 1:
            x = grid_to_bands(x, G, w, self.group)
        y = self.head(x)                                              # (b, Hb, W, d)
        if G == 1:
            return y
        y = y.contiguous()
        b, Hb, W, d = y.shape
        out = torch.empty((G * b, Hb, W, d), dtype=y.dtype, device=y.device)
        if y.is_cuda and dist.get_backend(self.group) == "gloo":
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, y.cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, y, group=self.group)
        return out.reshape(G, b, Hb, W, d).permute(1, 0, 2, 3, 4).reshape(b, G * Hb, W, d)


# ----------------------------------------------------------------------------------------------
# training: data-parallel gradient all-reduce
# ----------------------------------------------------------------------------------------------
class GradAllReducer(object):
    """Bucketed, backward-overlapped mean all-reduce of parameter gradients over RCCL: what DistributedDataParallel does for the
    reference (opv2v/opencood/tools/train_camera.py:105-110, train_utils / multi_gpu_utils init_distributed_mode), written out
    so that bucket size and stream are this package's choice rather than DDP's NVSwitch-era defaults.

    Parameters are packed in REVERSE registration order (backward reaches the last layers first) into flat fp32 buckets.  A
    post-accumulate hook marks a parameter ready; when a bucket is complete its gradients are copied into the flat buffer and
    ONE asynchronous all-reduce is issued for it, so the collective of the late layers runs under the backward of the early
    ones.  `finish()` (call after loss.backward(), before optimizer.step()) issues buckets that stayed incomplete, waits,
    divides by the world size and scatters the result back into p.grad.

    Parameters that receive no gradient (DDP's find_unused_parameters=True case: BevSegHead registers `static_head` although
    `target: dynamic` never runs it, bev_seg_head.py:17-33; DiscoNet's unused containers) must not stall the queue: every
    bucket carries one "used" word per parameter behind its gradients, so after the reduction all ranks know which
    parameters had a gradient on ANY rank.  Those that had none keep `p.grad = None` (the optimizer skips them, as under DDP)
    and a rank leaves the parameters it had no LOCAL gradient for out of its readiness count of the next step.  Should such a
    parameter receive a gradient after its bucket has gone out, one scalar all-reduce in `finish()` makes every rank re-reduce this step's buckets synchronously.

    Bucket size: xGMI is point-to-point, so a ring all-reduce moves 2 (N-1)/N of the bucket over each ~153 GB/s link in N-1 +
    N-1 steps of bucket / N bytes; at 8 GPUs a 32-MB bucket gives 4-MB chunks per step (bandwidth-bound, ~0.4 ms per bucket)
    while the whole CoBEVT model (~40 MB of fp32 gradients) still splits into two buckets, i.e. one of them overlaps with
    backward.  Tiny buckets would be latency-bound per ring step; one giant bucket would not overlap at all.
    """

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        self.buckets = []            # [dict(params, flat, offsets, n, pending, work, ready)]
        cur, cur_bytes = [], 0
        for p in reversed(params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._bucket_of = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._bucket_of[p] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.launched = []           # bucket indices in issue order (this step)
        self.launched_in_backward = 0    # buckets that went out before finish() in the last step (the overlap actually achieved)
        self.skip = set()            # id(p) of the parameters THIS rank had no gradient for in the previous step
        self._late = False
        self.enabled = True          # False: backward passes are local (gradient accumulation steps, like DDP.no_sync())

    def _close(self, plist):
        offs, n = [], 0
        for p in plist:
            offs.append(n)
            n += p.numel()
        flat = torch.zeros(n + len(plist), device=plist[0].device, dtype=torch.float32)     # gradients | one "used" word each
        self.buckets.append(dict(params=list(plist), flat=flat, offsets=offs, n=n, pending=len(plist), work=None, ready=set()))

    def _on_grad(self, p):
        if not self.enabled:
            return
        bi = self._bucket_of[p]
        b = self.buckets[bi]
        if id(p) in b["ready"]:
            return
        b["ready"].add(id(p))
        if bi in self.launched:       # only a parameter that was expected to stay unused can arrive after its bucket went out
            self._late = True
            return
        if id(p) not in self.skip:
            b["pending"] -= 1
        # buckets are issued strictly in order so that every rank posts the same sequence of collectives
        self._issue_ready()

    def _issue_ready(self, force=False):
        nxt = len(self.launched)
        while nxt < len(self.buckets) and (force or self.buckets[nxt]["pending"] <= 0):
            self._issue(nxt)
            nxt += 1

    def _pack(self, b):
        flat = b["flat"]
        used = []
        for p, o in zip(b["params"], b["offsets"]):
            view = flat[o:o + p.numel()]
            if p.grad is None:
                view.zero_()
                used.append(0.0)
            else:
                view.copy_(p.grad.reshape(-1))
                used.append(1.0)
        flat[b["n"]:].copy_(torch.tensor(used, dtype=torch.float32), non_blocking=True)

    def _issue(self, bi):
        b = self.buckets[bi]
        self._pack(b)
        if self.world > 1:
            b["work"] = torch.distributed.all_reduce(b["flat"], group=self.group, async_op=True)
        self.launched.append(bi)

    def finish(self):
        """after backward: issue what is left, wait, average, write back into p.grad"""
        self.launched_in_backward = len(self.launched)
        self._issue_ready(force=True)
        if self.world > 1:
            late = torch.tensor([1.0 if self._late else 0.0], device=self.buckets[0]["flat"].device)
            torch.distributed.all_reduce(late, group=self.group)
            if float(late.item()) > 0:            # some rank packed a bucket before one of its gradients existed: redo, in order
                for bi in self.launched:
                    b = self.buckets[bi]
                    if b["work"] is not None:
                        b["work"].wait()
                    self._pack(b)
                    b["work"] = torch.distributed.all_reduce(b["flat"], group=self.group, async_op=True)
        skip = set()
        for bi in self.launched:
            b = self.buckets[bi]
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            used = b["flat"][b["n"]:].tolist()
            if self.world > 1:
                b["flat"][:b["n"]].mul_(1.0 / self.world)
            pending = 0
            for p, o, u in zip(b["params"], b["offsets"], used):
                if p.grad is None:                # not expected locally next step either (readiness is a per-rank matter;
                    skip.add(id(p))               # the ORDER of the collectives is the same on every rank regardless)
                else:
                    pending += 1
                if u <= 0:                        # no gradient on any rank: leave p.grad as it is (None), as DDP does
                    continue
                g = b["flat"][o:o + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            b["pending"] = pending
            b["ready"] = set()
        self.skip = skip
        self.launched = []
        self._late = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
