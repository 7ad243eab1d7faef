"""Multi-GPU plumbing: independent proofs are partitioned over ranks (one process per GPU); the only
cross-rank traffic is the timing max-reduce and the gather of verdicts / proof bytes on rank 0.
No data-path collective exists for BASELINE configs 1-4.  Config 5 (one accumulated MSM over
the proofs of all GPUs) has exactly one exchange step: an all-gather of one 144-byte partial sum per rank."""
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
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_lists(local_list, dist=None):
    """Concatenate per-rank python lists in rank order on every rank."""
    if dist is None or not dist.is_initialized():
        return list(local_list)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local_list))
    return [x for part in out for x in part]


JAC_BYTES = 144


def all_gather_partials(partial_jac, n_invalid, dist=None, device="cpu"):
    """BASELINE config 5 exchange step: every rank contributes the partial sum of its share of the accumulated
    check (144-byte Jacobian point from `cpx_batch_verify_fused`) and its count of structurally invalid proofs.
    Returns (partials of all ranks concatenated in rank order, total invalid count) on every rank.
    One fixed-size uint8 all_gather = ncclAllGather over xGMI on GPUs (backend nccl is RCCL), gloo on CPU; the
    payload is 148 bytes per rank, so the step is latency-bound."""
    if len(partial_jac) != JAC_BYTES:
        raise ValueError("partial sum must be one 144-byte Jacobian point")
    if dist is None or not dist.is_initialized():   # (an initialised group of ONE rank still goes through the collective: the
        return bytes(partial_jac), int(n_invalid)    # single-GPU test of the RCCL path, tests/test_gpu_parity.py)
    import torch
    payload = bytes(partial_jac) + int(n_invalid).to_bytes(4, "little")
    mine = torch.tensor(list(payload), dtype=torch.uint8, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    raw = [bytes(t.cpu().tolist()) for t in parts]
    return b"".join(r[:JAC_BYTES] for r in raw), sum(int.from_bytes(r[JAC_BYTES:], "little") for r in raw)


def fused_verify_all_ranks(ctx, proofs, rand, dist=None, device="cpu"):
    """Config 5 on N GPUs: each rank runs the fused verifier over its loaded share, the partial sums are
    all-gathered and added locally on every rank (an EC addition is not an RCCL reduction op).  True iff every
    proof on every rank is valid."""
    part, bad = ctx.verify_batch_fused_partial(proofs, rand)
    parts, bad_total = all_gather_partials(part, bad, dist, device)
    return bad_total == 0 and ctx.sum_jac(parts)[1]
