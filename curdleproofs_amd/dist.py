"""Multi-GPU plumbing: independent proofs are partitioned over ranks (one process per GPU); the only
cross-rank traffic is the timing max-reduce and the gather of verdicts / proof bytes on rank 0.
No data-path collective exists for BASELINE configs 1-4."""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def partition(n_items, world, rank):
    """Contiguous block of item indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def max_over_ranks(value, dist=None, device="cpu"):
    """Max of a python float over all ranks (the bench contract's max-over-ranks step time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_lists(local_list, dist=None):
    """Concatenate per-rank python lists in rank order on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_list)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local_list))
    return [x for part in out for x in part]
