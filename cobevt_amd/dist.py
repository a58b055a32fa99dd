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
