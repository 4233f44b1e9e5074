"""One rank of a CPU rehearsal of bench.py's multi-rank path (started by tests/test_multiprocess.py through
bench.spawn_ranks): the launcher's environment, the gloo process group, the barrier-bracketed timed region, the
max over ranks and the contiguous frame shards of the strong-scaled workload -- with a sleep in place of the GPU step."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from orb_slam2_amd.dist import max_over_ranks, shard_frames, sum_over_ranks  # noqa: E402


def main():
    import torch.distributed as dist
    # STUB_BACKEND=nccl on a box without GPUs: the RCCL group cannot start and init_ranks must fall back to gloo
    rank, local_rank, world = bench.init_ranks(os.environ.get("STUB_BACKEND", "gloo"))
    assert bench.BACKEND["name"] == "gloo" and bench.reduce_device("cuda:0") is None
    assert local_rank == rank and world == int(os.environ["WORLD_SIZE"])
    total = 256
    mine = shard_frames(total, world, rank)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (1 + rank))  # the last rank is the slow one

    wall = bench.timed_steps(step, 5, 2, lambda: None, dist.barrier)
    wall_max = max_over_ranks(wall)
    assert len(calls) == 7
    covered = sum_over_ranks(len(mine))
    if rank == 0:
        line = bench.bench_line("stub", "frames/s", total * 5 / wall_max, world, 5, 2, wall_max, "strong", "u8", "none",
                                {"workload": "stub", "first": mine[0], "last": mine[-1], "covered": covered,
                                 "slowest_rank_s": wall_max, "this_rank_s": wall, "backend": bench.BACKEND["name"]})
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
