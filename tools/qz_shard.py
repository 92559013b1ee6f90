"""Block sharding across the GPUs of one node (no data-path collective).

The reference scales by independent blocks only: threads x CCtx each grabbing its own
accelerator instance, instances interleaved across physical devices
(/root/reference/src/qatseqprod.c:601-630, README.md:138).  Here the unit is the block and
the shard is a contiguous range of block indices per rank; results are gathered on the host
(each rank owns its output).  The only collectives are control-plane: a barrier and a MAX of
the elapsed time for the benchmark contract.
"""
from __future__ import annotations


def shard_range(n_blocks: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of the blocks rank `rank` owns: contiguous, balanced (sizes differ by <= 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_blocks, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def weak_offset(unit_bytes: int, world: int, rank: int) -> int:
    """weak scaling: every rank processes `unit_bytes`; ranks start at staggered offsets into
    the (repeated) corpus so that they do not all read the same bytes."""
    return rank * (unit_bytes // max(world, 1))


def reduce_max_seconds(seconds: float, dist=None, device=None) -> float:
    """MAX over ranks of a host-side duration (bench contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts, dist=None):
    """host-side gather of per-block sequence counts in block order (rank order == block order)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_counts)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local_counts))
    return [c for part in out for c in part]
