"""World-size-2 run of the multi-GPU bookkeeping on CPU (gloo): frame sharding,
barrier and max-over-ranks timing, exactly what bench.py does across GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from orb_slam2_amd.dist import max_over_ranks, shard_frames, sum_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = list(shard_frames(101, world, rank))
    dist.barrier()
    elapsed = 1.0 + rank  # rank 1 is the slow one
    tmax = max_over_ranks(elapsed)
    total = sum_over_ranks(len(frames))
    q.put((rank, frames[0], frames[-1], tmax, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_reduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, a0, b0, t0, n0), (r1, a1, b1, t1, n1) = res
    assert (a0, b0, a1, b1) == (0, 50, 51, 100)      # contiguous, disjoint, covering
    assert t0 == t1 == 2.0                            # the slowest rank's time everywhere
    assert n0 == n1 == 101.0                          # whole-job frame count


def test_bench_launcher_starts_ranks_and_relays_rank0():
    """bench.py --gpus N without torchrun: the launcher starts N rank processes with the torchrun environment and
    prints rank 0's line.  Driven here with 2 gloo ranks and a stub step on the CPU."""
    import json
    import bench
    rc, out = bench.spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "stub_bench_rank.py")], timeout=240)
    assert rc == 0, out
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2 and line["scaling"] == "strong"
    cfg = line["config"]
    assert (cfg["first"], cfg["last"], cfg["covered"]) == (0, 127, 256.0)
    # the slowest rank (0.02 s per step) sets the time; rank 0 alone would have needed half of it
    assert cfg["slowest_rank_s"] >= 5 * 0.02 and cfg["slowest_rank_s"] >= cfg["this_rank_s"]
    assert abs(line["value"] - 256 * 5 / cfg["slowest_rank_s"]) < 1e-6 * line["value"]
    assert abs(line["ms_per_step"] - cfg["slowest_rank_s"] / 5 * 1e3) < 1e-9


def test_bench_ranks_fall_back_to_gloo_when_rccl_cannot_start():
    """The ranks ask for the nccl (RCCL) backend on a machine where it cannot start (no GPU here): every rank must fall
    back to gloo, report so, and the timed region, the max over ranks and rank 0's line must come out as usual."""
    import json
    import bench
    rc, out = bench.spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "stub_bench_rank.py")], extra_env={"STUB_BACKEND": "nccl"},
                                timeout=300)
    assert rc == 0, out
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["backend"] == "gloo" and line["config"]["covered"] == 256.0


def test_bench_launcher_reports_a_failing_rank():
    import bench
    rc, _ = bench.spawn_ranks(2, [sys.executable, "-c", "import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)"], timeout=60)
    assert rc == 3
