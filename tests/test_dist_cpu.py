"""CPU, world sizes 2 / 4 / 5 / 8, gloo: the agent-sharded multi-GPU path (cobevt_amd/dist.py) — task dealing, the single
all-gather, per-frame re-assembly — with the ORACLE injected as the compute (the HIP kernels need a GPU; the
exchange logic is device independent).  Each rank's fused output must equal a single-process run of its frame."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from cobevt_amd import dist as cdist
from cobevt_amd import host, synth


def test_task_dealing_is_a_bijection():
    for world in (1, 2, 4, 5, 8):
        for agents in (2, 5, 8):
            seen = {}
            for r in range(world):
                for slot, (f, a) in enumerate(cdist.tasks_of_rank(r, world, agents)):
                    assert (f, a) not in seen and 0 <= f < world and 0 <= a < agents
                    seen[(f, a)] = r * agents + slot
            assert len(seen) == world * agents
            for f in range(world):
                assert cdist.gather_index(f, world, agents) == [seen[(f, a)] for a in range(agents)]


class _OracleModel(object):
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def encode_agents(self, batch):
        import oracle.corpbevt as o
        return o.encode_agents(self.sd, self.cfg, batch)

    def fuse_and_decode(self, feats, tm, record_len):
        import oracle.corpbevt as o
        return o.fuse_and_decode(self.sd, self.cfg, feats, tm, record_len)


def _frame(f, agents):
    return synth.opv2v_batch(agents=agents, cams=2, image=128, max_cav=3, seed=f)


def _worker(rank, world, port, agents, ret):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(1)
    torch.set_grad_enabled(False)
    r, w, _ = cdist.init_from_env("gloo")
    import copy
    cfg = synth.corpbevt_small_config()
    sd = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).state_dict()
    model = _OracleModel(sd, cfg)
    frames = [_frame(f, agents) for f in range(world)]
    tasks = cdist.tasks_of_rank(r, w, agents)
    task_batch = {k: torch.stack([frames[f][k][a] for f, a in tasks]) for k in ("inputs", "intrinsic", "extrinsic")}
    pipe = cdist.AgentShardedCoBEVT(model, r, w, agents)
    out = pipe.step(task_batch, frames[r]["transformation_matrix"], frames[r]["record_len"])["dynamic_seg"]
    import oracle.corpbevt as o
    ref = o.corpbevt_forward(sd, cfg, frames[r])["dynamic_seg"]
    ret[rank] = float((out - ref).abs().max())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    ret = mp.Manager().dict()
    mp.spawn(fn, args=(world, _free_port()) + tuple(args) + (ret,), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))
    return dict(ret)


@pytest.mark.parametrize("world,agents", [(2, 2), (5, 2), (8, 3)])
def test_agent_sharded_pipeline_gloo(world, agents):
    """weak scaling: `world` frames in flight, world * agents tasks dealt round-robin, one all-gather"""
    for r, err in _spawn(_worker, world, agents).items():
        assert err <= 1e-5, "rank %d: sharded output differs from the single-process frame by %.3e" % (r, err)


def test_strong_scaling_index_bookkeeping():
    for world in (1, 2, 4, 5, 8):
        for agents in (1, 2, 5, 8):
            s = cdist.slots_per_rank(world, agents)
            owners = {}
            for r in range(world):
                ids = cdist.agents_of_rank(r, world, agents)
                assert len(ids) <= s
                for slot, a in enumerate(ids):
                    owners[a] = r * s + slot
            assert sorted(owners) == list(range(agents))
            assert cdist.strong_gather_index(world, agents) == [owners[a] for a in range(agents)]


def _strong_worker(rank, world, port, agents, ret):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(1)
    torch.set_grad_enabled(False)
    r, w, _ = cdist.init_from_env("gloo")
    import copy
    import oracle.corpbevt as o
    cfg = synth.corpbevt_small_config()
    sd = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).state_dict()
    frame = _frame(7, agents)
    out = cdist.FrameShardedCoBEVT(_OracleModel(sd, cfg), r, w, agents).step(frame)["dynamic_seg"]
    ref = o.corpbevt_forward(sd, cfg, frame)["dynamic_seg"]
    ret[rank] = float((out - ref).abs().max())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,agents", [(2, 3), (5, 3), (8, 3)])
def test_frame_sharded_strong_scaling_gloo(world, agents):
    """strong scaling: ONE frame, rank r encodes agents r, r + G, ..; all-gather (surplus ranks send zero blocks); the fusion
    is replicated, so EVERY rank must reproduce the single-process frame"""
    for r, err in _spawn(_strong_worker, world, agents).items():
        assert err <= 1e-5, "rank %d: frame-sharded output differs from the single-process frame by %.3e" % (r, err)


LIDAR_SMALL = dict(input_dim=32, mlp_dim=64, agent_size=4, window_size=4, dim_head=32, drop_out=0.1, depth=2, mask=True)


def _lidar_small_inputs():
    x = synth.procedural_input("lidar.small.x", (1, 4, 32, 16, 16), 3)
    mask = torch.ones(1, 16, 16, 1, 4)
    mask[0, :, :, :, 3] = 0
    mask[0, :5, 9:, :, 1] = 0
    return x, mask


def _lidar_worker(rank, world, port, ret):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(1)
    torch.set_grad_enabled(False)
    r, w, _ = cdist.init_from_env("gloo")
    import oracle.swap_fusion as o
    args = LIDAR_SMALL
    sd = synth.fill_module_(host.SwapFusionEncoder(dict(args)), 0).state_dict()
    x, mask = _lidar_small_inputs()
    ref = o.swap_fusion_encoder(sd, "", args, x, mask)                      # (b d h w)
    L, win, dh = args["agent_size"], args["window_size"], args["dim_head"]
    stages = []
    for i in range(args["depth"]):
        n = o.block_names("", i, True)
        for mode, (ap, fp) in ((0, (n[0], n[1])), (1, (n[2], n[3]))):
            stages.append((mode, lambda xl, m, ap=ap, fp=fp, mode=mode: o.swap_stage(sd, ap, fp, xl, m, mode, dh, L, win)))
    pipe = cdist.RowShardedFuseBEVT(stages, lambda xl: o.mlp_head(sd, "", xl), r, w, win)
    per = L // w
    mine = x[:, r * per:(r + 1) * per].permute(0, 1, 3, 4, 2).contiguous()     # this rank's agents, channels-last
    out = pipe.step(mine, mask)                                              # (b h w d) on every rank
    ret[rank] = float((out.permute(0, 3, 1, 2) - ref).abs().max())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_row_shard_layout_roundtrip_single_process():
    """band <-> grid re-partition is the identity at world 1 and the masks slice consistently"""
    x, mask = _lidar_small_inputs()
    xl = x.permute(0, 1, 3, 4, 2).contiguous()
    assert cdist.bands_to_grid(xl, 1, 4) is xl and cdist.grid_to_bands(xl, 1, 4) is xl
    for world in (2, 4):
        rows = []
        for r in range(world):
            mg = cdist.mask_grid(mask, r, world, 4)
            X = 16 // 4
            want = torch.stack([mask[:, i * X + xx] for i in range(4) for xx in range(r * X // world, (r + 1) * X // world)], 1)
            assert torch.equal(mg, want)
            rows.append(cdist.mask_band(mask, r, world))
        assert torch.equal(torch.cat(rows, 1), mask)
    with pytest.raises(ValueError):
        cdist.shard_check(16, 4, 8, 4)


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_fusebevt_gloo(world):
    """LiDAR-style SwapFusionEncoder over row-sharded maps: agent->band all-to-all, window passes on bands, grid passes on the
    (i, x_local) re-ordered rows (one all-to-all each way per block), all-gather of the fused bands == the single-process run"""
    for r, err in _spawn(_lidar_worker, world).items():
        assert err <= 1e-5, "rank %d: row-sharded FuseBEVT differs from the single-process result by %.3e" % (r, err)


def _grad_worker(rank, world, port, bucket_bytes, ret):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(1)
    cdist.init_from_env("gloo")
    torch.manual_seed(0)                                   # same initial weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.LayerNorm(64),
                              torch.nn.Linear(64, 8))
    unused = torch.nn.Parameter(torch.ones(5))             # a parameter that gets no gradient on rank 0
    never = torch.nn.Parameter(torch.ones(3))              # ... and one no rank ever uses, registered last (BevSegHead.static_head
    params = list(net.parameters()) + [unused, never]      #     under `target: dynamic`): it sits in bucket 0 of the reversed order
    red = cdist.GradAllReducer(params, bucket_bytes=bucket_bytes)
    worst, early = 0.0, []
    for step in range(3):
        xs = [torch.randn(4, 16, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world)]
        # reference: every rank's gradient computed locally (reducer switched off), averaged
        red.enabled = False
        ref = [torch.zeros_like(p) for p in params]
        for r in range(world):
            for p in params:
                p.grad = None
            (net(xs[r]).square().mean() + (unused.sum() if r == 1 else 0.0)).backward()
            for a, p in zip(ref, params):
                if p.grad is not None:
                    a += p.grad / world
        red.enabled = True
        for p in params:
            p.grad = None
        loss = net(xs[rank]).square().mean() + (unused.sum() if rank == 1 else 0.0)
        loss.backward()
        red.finish()
        assert never.grad is None                          # DDP (find_unused_parameters) leaves it None: the optimizer skips it
        got = [p.grad.clone() for p in params if p is not never]
        worst = max(worst, max(float((a - b).abs().max()) for a, b in zip(got, [r_ for r_, p in zip(ref, params) if p is not never])))
        early.append(red.launched_in_backward)
    ret[rank] = (worst, len(red.buckets), early)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,bucket_bytes", [(2, 32 << 20), (2, 4096), (4, 1024)])
def test_grad_all_reducer_matches_averaged_gradients(world, bucket_bytes):
    """train_camera.py:105-110 (DistributedDataParallel) equivalent: one bucket, several buckets, a parameter without a gradient
    on some ranks, two consecutive steps"""
    ret = _spawn(_grad_worker, world, bucket_bytes)
    for r in range(world):
        assert ret[r][0] <= 1e-6, ret
    assert ret[0][1] == 1 if bucket_bytes > (1 << 20) else ret[0][1] > 1
    if ret[0][1] > 1:
        # from the second step on the never-used parameter no longer holds bucket 0 back: buckets go out DURING backward
        for r in range(world):
            assert ret[r][2][0] == 0 and ret[r][2][1] > 0 and ret[r][2][2] > 0, ret
