"""Worker of tests/test_dist_gpu.py: one rank of a multi-process exchange test.  Launched as
`python tests/peer_worker.py <case>` with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment (torchrun
convention).  Several ranks may share one GPU (COBEVT_DIST_BACKEND=gloo; the direct peer-window exchange works between
processes on one device as well as across xGMI) - that is how a one-GPU box covers the N > 1 code."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cobevt_amd import dist as cdist  # noqa: E402


def block(rank, j, shape, step, dev, dtype):
    g = torch.Generator().manual_seed(1000 * step + 16 * rank + j)
    return torch.randn(shape, generator=g).to(device=dev, dtype=dtype)


def case_direct(rank, world, dev):
    """strong plan (all-gather, slot = agent id) and weak plan (task -> frame owner) against the blocks every rank can
    recompute, several exchanges in a row (window reuse = the ack handshake), also replayed from a captured HIP graph"""
    agents, shape, dtype = 5, (8, 8, 16), torch.bfloat16
    spin = 2000000
    # --- strong ---
    ex = cdist.DirectExchange(shape, dtype, agents, rank, world, spin_limit=spin)
    ex.plan(*cdist.direct_plan_strong(rank, world, agents))
    mine = cdist.agents_of_rank(rank, world, agents)
    for step in range(4):
        local = torch.stack([block(rank, j, shape, step, dev, dtype) for j in range(len(mine))]) if mine else \
            torch.empty((0,) + shape, device=dev, dtype=dtype)
        got = ex(local).clone()
        st, ep = ex.status()
        assert st == 0 and ep == step + 1, (st, ep)
        for a in range(agents):
            r, j = a % world, a // world
            assert torch.equal(got[a], block(r, j, shape, step, dev, dtype)), ("strong", step, a)
    # --- captured ---
    local = torch.zeros((max(1, len(mine)),) + shape, device=dev, dtype=dtype)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ex(local)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        win = ex(local)
        out = win.float().sum(dim=(1, 2, 3))
    for step in range(10, 13):
        for j in range(len(mine)):
            local[j].copy_(block(rank, j, shape, step, dev, dtype))
        g.replay()
        torch.cuda.synchronize()
        want = torch.stack([block(a % world, a // world, shape, step, dev, dtype).float().sum() for a in range(agents)])
        assert torch.allclose(out, want, rtol=1e-5, atol=1e-4), ("graph", step)
    assert ex.status()[0] == 0
    ex.close()
    # --- weak ---
    ex = cdist.DirectExchange(shape, dtype, agents, rank, world, spin_limit=spin)
    ex.plan(*cdist.direct_plan_weak(rank, world, agents))
    for step in range(3):
        local = torch.stack([block(rank, j, shape, step, dev, dtype) for j in range(agents)])
        got = ex(local).clone()
        assert ex.status()[0] == 0
        for a in range(agents):
            t = rank * agents + a                        # frame `rank`, agent a = task t -> encoded by rank t % world, slot t // world
            assert torch.equal(got[a], block(t % world, t // world, shape, step, dev, dtype)), ("weak", step, a)
    ex.close()


def case_collective(rank, world, dev):
    """the RCCL (or gloo dry-run) exchange of cobevt_amd.dist against the same recomputable blocks"""
    agents, shape, dtype = 5, (8, 8, 16), torch.bfloat16
    for step in range(3):
        local = torch.stack([block(rank, j, shape, step, dev, dtype) for j in range(agents)])
        got = cdist.exchange_features(local, rank, world, agents)
        for a in range(agents):
            t = rank * agents + a
            assert torch.equal(got[a], block(t % world, t // world, shape, step, dev, dtype)), ("weak", step, a)
        mine = cdist.agents_of_rank(rank, world, agents)
        got = cdist.exchange_features_strong(local, len(mine), rank, world, agents)
        for a in range(agents):
            assert torch.equal(got[a], block(a % world, a // world, shape, step, dev, dtype)), ("strong", step, a)


def main():
    case = sys.argv[1]
    rank, world, local_rank = cdist.init_from_env()
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_grad_enabled(False)
    {"direct": case_direct, "collective": case_collective}[case](rank, world, dev)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    print("PEER_WORKER_OK rank %d/%d case %s backend %s" % (rank, world, case, os.environ.get("COBEVT_DIST_BACKEND", "nccl")), flush=True)


if __name__ == "__main__":
    main()
