"""N > 1 path on CPU: two processes, gloo backend.  Checks the block sharding (every block
owned exactly once, in order), the host-side gather and the MAX-over-ranks timing reduction
bench.py uses.  The per-block work here is the CPU oracle standing in for a rank's GPU
(tests may use the oracle as the checker); on the GPU box bench.py runs the HIP path."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_blocks, q):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch.distributed as dist
    import qz_bind as B
    import qz_corpus as K
    import qz_shard as S

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = K.mix(2, n_blocks * 32768)
    lo, hi = S.shard_range(n_blocks, world, rank)
    orc = B.Oracle()
    counts = []
    for b in range(lo, hi):
        blk = data[b * 32768:(b + 1) * 32768]
        n, _ = orc.find(orc.profile(1, len(blk)), blk)
        counts.append(int(n))
    allc = S.gather_counts(counts, dist)
    tmax = S.reduce_max_seconds(0.25 * (rank + 1), dist)
    dist.barrier()
    if rank == 0:
        q.put((allc, tmax, (lo, hi)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_blocks", [7, 8])
def test_two_rank_sharding_and_gather(n_blocks):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import qz_bind as B
    import qz_corpus as K

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_blocks, q)) for r in range(2)]
    [p.start() for p in procs]
    allc, tmax, _ = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process truth
    data = K.mix(2, n_blocks * 32768)
    orc = B.Oracle()
    want = [int(orc.find(orc.profile(1, 32768), data[b * 32768:(b + 1) * 32768])[0]) for b in range(n_blocks)]
    assert allc == want
    assert tmax == pytest.approx(0.5)  # MAX over ranks, not the sum or the mean


def test_shard_ranges_partition_exactly():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import qz_shard as S
    for n in (0, 1, 7, 8, 8192, 524288, 524289):
        for world in (1, 2, 4, 8):
            spans = [S.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S.shard_range(8, 2, 2)
