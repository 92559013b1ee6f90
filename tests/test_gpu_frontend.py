"""GPU: the batch front-end (include/qzstd_frontend.h) over the REAL library — the leg bench.py's headline
(`value`: input MB/s through ZSTD_compress2) runs through.  The reference shape it stands in for is the per-chunk ZSTD_compress2
loop of /root/reference/test/benchmark.c:300-321, one frame per chunk, many threads.  Frames must be byte-identical to the frames
libzstd builds from the ORACLE's sequences, every block must come from an announcement, no callback may fail (round-3 verdict:
until now the component was compared with the oracle over the mock only)."""
import ctypes as C

import pytest

import qz_bind as B
import qz_corpus as K

pytestmark = pytest.mark.gpu


def oracle_frames(zstd, oracle, data, chunk, level, ext_rep=0):
    zo = zstd.cctx(level, producer=oracle.producer_addr, state=None, fallback=False, validate=True,
                   **({"ext_repcodes": ext_rep} if ext_rep else {}))
    _, frames = zstd.compress_chunks(zo, data, chunk)
    zstd.free(zo)
    return frames


@pytest.mark.parametrize("level,chunk,threads,blocks", [(1, 131072, 8, 96), (6, 131072, 8, 40), (12, 32768, 6, 64), (3, 65536, 5, 50)])
def test_front_end_frames_equal_the_oracles(gpu_plugin, zstd, oracle, level, chunk, threads, blocks):
    front = B.Front()
    data = K.by_name("system", blocks * chunk + 4321, seed=level + 60)  # a ragged last chunk
    n = (len(data) + chunk - 1) // chunk
    frames, st, fs = front.frames(data, chunk, level, threads, segment=8 * chunk, jobs=2)
    assert len(frames) == n
    want = oracle_frames(zstd, oracle, data, chunk, level)
    bad = [c for c in range(n) if frames[c] != want[c]]
    assert not bad, "frames %s differ from libzstd + oracle (level %d, chunk %d)" % (bad[:8], level, chunk)
    assert st[0] == 2 * n and st[1] == 0, "not every block came from an announcement: %s" % st
    assert fs[0] == 0, "producer callbacks failed: %s" % fs
    # and the reference's own criterion on top (test/benchmark.c:329-339): every frame decodes to its chunk
    for c in (0, n // 2, n - 1):
        blk = data[c * chunk:(c + 1) * chunk]
        assert zstd.decompress(frames[c], len(blk)) == blk


def test_front_end_one_thread_and_default_segment(gpu_plugin, zstd, oracle):
    """segmentBytes = 0 (2 MiB), one worker: the degenerate pool; same frames"""
    front = B.Front()
    data = K.by_name("mix", 3 * (1 << 20) + 17, seed=66)
    frames, st, fs = front.frames(data, 131072, 1, 1)
    assert frames == oracle_frames(zstd, oracle, data, 131072, 1)
    assert st[1] == 0 and fs[0] == 0, (st, fs)


@pytest.mark.parametrize("level,chunk,blocks", [(1, 131072, 1024), (3, 131072, 512), (6, 131072, 192), (12, 32768, 768)])
def test_front_end_under_load_every_frame_is_the_oracles(gpu_plugin, zstd, oracle, level, chunk, blocks):
    """round 4: the announcements' completion is the blocks' count words in pinned memory (no stream query), their staging copy a copy kernel,
    a state keeps up to four announcements and the front-end two or three claims announced ahead — all of it under the load the bench runs it
    at: 18 workers on every core, hundreds of launches in flight over 16 hardware queues, three jobs back to back over the same buffers (result
    areas reused while later launches write theirs).  Every frame of every job has to be libzstd's frame from the ORACLE's sequences: a count
    word that overtook its entries, or entries of the previous job, would show here and nowhere else"""
    front = B.Front()
    data = K.by_name("system", blocks * chunk - 12345, seed=level + 90)
    n = (len(data) + chunk - 1) // chunk
    want = oracle_frames(zstd, oracle, data, chunk, level)
    for threads, seg in ((18, 0), (7, 4 * chunk)):
        def check(job, frames):
            bad = [c for c in range(n) if frames[c] != want[c]]
            assert not bad, "job %d: frames %s differ from libzstd + oracle (level %d, %d threads)" % (job, bad[:8], level, threads)

        frames, st, fs = front.frames(data, chunk, level, threads, segment=seg, jobs=3, each=check)
        assert st[0] == 3 * n and st[1] == 0, "not every block came from an announcement: %s" % (st,)
        assert fs[0] == 0, "producer callbacks failed: %s" % (fs,)
