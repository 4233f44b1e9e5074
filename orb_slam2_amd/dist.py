"""Frame-level data parallelism helpers (one process per GPU, no data-path collective).

Frames are independent units of work, so multi-GPU runs shard frame indices
across ranks; torch.distributed is used only to line ranks up (barrier) and to
take the slowest rank's time.
"""
import os


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_frames(n_frames, world, rank):
    """Contiguous chunk of frame indices for `rank` (so frame k and k+1 stay on one GPU for matching).

    The first n_frames % world ranks get one extra frame.
    """
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def max_over_ranks(value, device=None):
    """MAX-all-reduce of a python float when a process group is up, else the value itself."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
