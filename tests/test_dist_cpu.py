"""CPU, world size 2, gloo: the agent-sharded multi-GPU path (cobevt_amd/dist.py) — task dealing, the single
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
    torch.set_num_threads(2)
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


def test_agent_sharded_pipeline_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, 2, ret), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for r, err in ret.items():
        assert err <= 1e-5, "rank %d: sharded output differs from the single-process frame by %.3e" % (r, err)
