"""GPU tests of the multi-rank code paths (cobevt_amd/dist.py, csrc/peer_gather.hip, bench.py's own launcher).

A one-GPU box cannot host two RCCL ranks, so there the ranks share GPU 0: the direct peer-window exchange runs for real
(hipIpc between processes, system-scope flags), the torch.distributed legs fall back to the gloo dry-run backend.  With
>= 2 GPUs the same workers run one rank per GPU over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(case, world, backend, timeout=240):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if backend == "gloo":
            env["COBEVT_DIST_BACKEND"] = "gloo"
        else:
            env.pop("COBEVT_DIST_BACKEND", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_worker.py"), case], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "PEER_WORKER_OK" in o, "rank %d failed:\n%s" % (r, o[-3000:])


def test_direct_exchange_single_rank(cuda):
    """world 1: the window is its own peer (kernels, flags, the __cuda_array_interface__ view)"""
    from cobevt_amd import dist as cdist
    ex = cdist.DirectExchange((4, 4, 8), torch.float32, 3, 0, 1)
    ex.plan([-1, 0], [2, 0])
    local = torch.arange(2 * 128, device=cuda, dtype=torch.float32).reshape(2, 4, 4, 8)
    for step in range(3):
        win = ex(local + step)
        torch.cuda.synchronize()
        assert torch.equal(win[2], local[0] + step) and torch.equal(win[0], local[1] + step)
    assert ex.status() == (0, 3)
    ex.close()


@pytest.mark.parametrize("world", [2, 4])
def test_direct_exchange_between_processes(cuda, world):
    """hipIpc windows between `world` processes (one per GPU when the box has them, otherwise sharing GPU 0)"""
    _launch("direct", world, "nccl" if torch.cuda.device_count() >= world else "gloo")


def test_collective_exchange_gloo_dry_run(cuda):
    _launch("collective", 2, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (>= 2 GPUs)")
def test_collective_exchange_rccl(cuda):
    _launch("collective", 2, "nccl")


def _bench(extra, env=None, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-roofline",
           "--no-extra"] + extra
    out = subprocess.run(cmd, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks(cuda):
    """`python bench.py --gpus 2` with no launcher starts 2 ranks itself; every rank checks its sharded frame against the
    single-process forward before timing (bench.py raises otherwise).  RCCL with >= 2 GPUs, gloo dry-run on one.  Layout of the
    line at N > 1: headline = throughput mode (the N = 1 quantity), `latency_mode` with frame_latency_ms beside it,
    the N = 1 reference of the same box and the efficiencies derived from it"""
    two = torch.cuda.device_count() >= 2
    res = _bench(["--gpus", "2"], env=None if two else {"COBEVT_DIST_BACKEND": "gloo"})
    assert res["n_gpus"] == 2 and res["rccl_ranks"]["world_size"] == 2
    assert res["rccl_ranks"]["backend"] == ("nccl" if two else "gloo")
    if two:
        assert res["rccl_ranks"]["distinct_gpus"] == 2
    assert res["mode"] == "throughput" and res["scaling"] == "weak" and res["value"] > 0
    lm = res["latency_mode"]
    assert lm["mode"] == "latency" and lm["scaling"] == "strong" and lm["value"] > 0 and lm["frame_latency_ms"] > 0, lm
    assert res["latency_mode_unpipelined"]["frame_latency_ms"] > 0
    assert res["single_gpu_reference"]["frames_per_sec"] > 0, res["single_gpu_reference"]
    sc = res["scaling_summary"]
    assert sc["throughput_scaling_efficiency_vs_n1"] > 0 and sc["frame_latency_speedup_vs_n1"] > 0
    assert res["box_calibration"]["mfma_bf16_tflops"] > 0
    # the direct peer-window legs come from the isolated second job (bench.py --direct-probe) and are merged into the one line
    assert "direct_peer_write" in res["all_gather_us"] and "rccl_all_gather_into_tensor" in res["all_gather_us"]
    assert res["latency_mode_direct_gather"].get("value", 0) > 0, res["latency_mode_direct_gather"]


@pytest.mark.parametrize("world", [5, 8])
def test_bench_agent_per_gpu_partitions(cuda, world):
    """BASELINE configs[3] (5 agents on 5 GPUs, one each) and the 8-GPU node with 5 agents (three surplus ranks that encode nothing
    and still take part in the exchange and the replicated fusion): bench.py's latency mode starting its own ranks - every rank
    checks its output against the single-process forward of the whole frame before timing.  RCCL when the box has the GPUs,
    otherwise gloo ranks sharing GPU 0 (the HIP kernels and the partition logic are the same; only the collective differs)"""
    real = torch.cuda.device_count() >= world
    res = _bench(["--gpus", str(world), "--mode", "latency"], env=None if real else {"COBEVT_DIST_BACKEND": "gloo"}, timeout=1200)
    assert res["n_gpus"] == world and res["rccl_ranks"]["world_size"] == world
    assert res["mode"] == "latency" and res["scaling"] == "strong" and res["value"] > 0 and res["frame_latency_ms"] > 0
    assert "agent a on GPU a mod %d" % world in res["config"]["parallelism"]
    assert res["single_gpu_reference"]["frames_per_sec"] > 0
