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


def parse_cpulist(text: str) -> list[int]:
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]; anything unparsable -> []"""
    out = []
    try:
        for part in text.strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
    except ValueError:
        return []
    return sorted(set(out))


def plan_rank_cpus(rank: int, node_of_rank: list[int], cpus_of_node: dict[int, list[int]], allowed: list[int]) -> list[int]:
    """CPUs for rank `rank` of a one-process-per-GPU job on one host: the CPUs of the NUMA node its GPU hangs off (reference: DMA memory on the
    instance's node, src/qatseqprod.c:216-246 — here the threads that copy into that memory go there too), restricted to what the process
    may use, shared out evenly between the ranks whose GPUs sit on the same node (the k-th such rank takes every n-th CPU from k on, so
    hyperthread siblings — usually numbered far apart — stay together as the kernel lists them).  [] = leave the affinity alone: unknown
    node, no CPU list, or fewer than two CPUs per rank to hand out."""
    if not (0 <= rank < len(node_of_rank)):
        return []
    node = node_of_rank[rank]
    if node is None or node < 0 or node not in cpus_of_node:
        return []
    ok = set(allowed)
    cpus = [c for c in cpus_of_node[node] if c in ok]
    mates = [r for r, n in enumerate(node_of_rank) if n == node]
    if len(cpus) < 2 * len(mates):
        return []
    per = len(cpus) // len(mates)
    k = mates.index(rank)
    return cpus[k * per:(k + 1) * per]
