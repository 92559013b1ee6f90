"""Host logic on CPU: qat-zstd-plugin_amd/host/qatseqprod.c linked against tests/mock/mock_hip.c (malloc as device
memory, synchronous streams, the oracle as the kernel) instead of the HIP layer.  Covers what otherwise only runs on
a GPU box: the cross-thread coalescer, announced and guessed look-ahead (incl. stale guesses and an unreadable page
behind the buffer), callbacks spanning several grid blocks, the per-slot mode, chain levels' workspace plumbing.
The mock exists under tests/ only; the product library has no CPU path (test_host_cpu.py checks that)."""
import ctypes as C
import mmap
import os
import subprocess
import threading

import pytest

import qz_bind as B
import qz_corpus as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_SO = os.path.join(ROOT, "tests", "mock", "libqatseqprod_mock.so")


def build_shared(cmd, out):
    """gcc ... -o <out>: built aside and renamed, so that pytest-xdist workers running the same fixture never load a half-written file"""
    tmp = "%s.%d.tmp" % (out, os.getpid())
    subprocess.check_call([tmp if x == out else x for x in cmd])
    os.replace(tmp, out)


@pytest.fixture(scope="module")
def mock(oracle):
    srcs = [os.path.join(B.PKG_DIR, "host", "qatseqprod.c"), os.path.join(B.PKG_DIR, "csrc", "qzstd_profile.c"),
            os.path.join(ROOT, "tests", "mock", "mock_hip.c"), os.path.join(ROOT, "oracle", "qzstd_oracle.c")]
    build_shared(["gcc", "-O2", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-DQZ_TEST_HOOKS", "-shared", "-fPIC", "-pthread",
                  "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-o", MOCK_SO] + srcs, MOCK_SO)
    plug = B.Plugin(MOCK_SO)
    assert plug.lib.QZSTD_startQatDevice() == 0
    yield plug
    plug.lib.QZSTD_stopQatDevice()


import contextlib


@contextlib.contextmanager
def restarted(mock, **env):
    """stop the (mock) device, restart it with environment variables set, and put everything back afterwards"""
    mock.lib.QZSTD_stopQatDevice()
    os.environ.update(env)
    try:
        assert mock.lib.QZSTD_startQatDevice() == 0
        yield
    finally:
        for k in env:
            del os.environ[k]
        mock.lib.QZSTD_stopQatDevice()
        assert mock.lib.QZSTD_startQatDevice() == 0


def frames_of(zstd, producer_addr, state, addr, total, chunk, level, before=None, order=None, **params):
    zc = zstd.cctx(level, producer=producer_addr, state=state, fallback=False, validate=True, **params)
    cap = zstd.lib.ZSTD_compressBound(chunk)
    dst = C.create_string_buffer(cap)
    out = {}
    idx = list(range((total + chunk - 1) // chunk))
    for c in (order or idx):
        if before:
            before(c)
        r = zstd.lib.ZSTD_compress2(zc, dst, cap, C.c_void_p(addr + c * chunk), min(chunk, total - c * chunk))
        assert not zstd.is_error(r), zstd.err(r)
        out[c] = dst.raw[:r]
    zstd.free(zc)
    return [out[c] for c in idx]


def oracle_frames(zstd, oracle, data, chunk, level):
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    return frames_of(zstd, oracle.producer_addr, None, C.addressof(buf), len(data), chunk, level)


def stats_of(plug, st):
    s = (C.c_ulong * 4)()
    plug.lib.QZSTD_hintStats(st, C.byref(s))
    return list(s)


@pytest.mark.parametrize("level,chunk", [(1, 131072), (3, 65536), (6, 131072), (12, 32768)])
def test_unchanged_caller_reads_nothing_but_the_callbacks_block(mock, zstd, oracle, level, chunk):
    """the library touches [src, src + srcSize) of a callback and nothing else: the opt-in transparent look-ahead of rounds 1-4
    (QZSTD_HIP_LOOKAHEAD) is gone (round-4 verdict, item 6), the variable is ignored.  The page behind the buffer is unreadable:
    a read behind the last block would fault."""
    libc = C.CDLL(None, use_errno=True)
    page = mmap.PAGESIZE
    data = K.by_name("mix", 24 * chunk, seed=level)
    mm = mmap.mmap(-1, len(data) + page)
    base = C.addressof(C.c_char.from_buffer(mm))
    mm[:len(data)] = data
    assert libc.mprotect(C.c_void_p(base + len(data)), C.c_size_t(page), 0) == 0
    want = oracle_frames(zstd, oracle, data, chunk, level)
    for env in ({}, {"QZSTD_HIP_LOOKAHEAD": "1"}, {"QZSTD_HIP_LOOKAHEAD": "2"}):
        with restarted(mock, **env):
            st = mock.lib.QZSTD_createSeqProdState()
            got = frames_of(zstd, mock.producer_addr, st, base, len(data), chunk, level)
            stats = stats_of(mock, st)
            mock.lib.QZSTD_freeSeqProdState(st)
        assert got == want
        assert stats[0] == 0 and stats[1] == 24 and stats[2] == 0, (env, stats)  # every block took the per-block path, nothing announced
    assert libc.mprotect(C.c_void_p(base + len(data)), C.c_size_t(page), 3) == 0
    del got


def test_announcements_double_buffered_and_grid_spanning(mock, zstd, oracle):
    data = K.by_name("system", 16 * 131072)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    st = mock.lib.QZSTD_createSeqProdState()
    seg = 4 * 131072
    assert mock.lib.QZSTD_hintSource(st, buf, seg, 131072, 3) == 0

    def ahead(c):
        if c % 4 == 0 and (c + 4) * 131072 < len(data):
            assert mock.lib.QZSTD_hintSource(st, C.byref(buf, (c + 4) * 131072), seg, 131072, 3) == 0

    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), 131072, 3, before=ahead)
    assert stats_of(mock, st)[:3] == [16, 0, 4]
    assert got == oracle_frames(zstd, oracle, data, 131072, 3)
    # a 64 KiB grid serves 128 KiB callbacks: two independently parsed halves joined
    assert mock.lib.QZSTD_hintSource(st, buf, 8 * 131072, 65536, 1) == 0
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), 8 * 131072, 131072, 1)
    assert stats_of(mock, st)[0] == 24
    mock.lib.QZSTD_freeSeqProdState(st)
    assert b"".join(zstd.decompress(f, 131072) for f in got) == data[:8 * 131072]
    assert mock.lib.QZSTD_hintSource(None, buf, 1000, 131072, 1) == -1


def test_four_announcements_ahead_and_the_stable_flag(mock, zstd, oracle):
    """round 4: a state holds FOUR announcements (a caller may run three segments ahead: the batch front-end keeps two announced beyond the
    one it is entropy-coding), and QZSTD_hintSourceEx(..., QZSTD_HINT_STABLE) serves by address without the per-callback memcmp.  Frames are the
    oracle's either way; unknown flag bits are refused; without the flag a rewritten buffer is still caught (and with it, by contract, not)"""
    L = mock.lib
    L.QZSTD_hintSourceEx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]
    data = K.by_name("system", 24 * 131072, seed=11)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    seg = 4 * 131072
    for stable in (0, 1):
        st = L.QZSTD_createSeqProdState()
        for k in range(3):  # three segments ahead of the first callback
            assert L.QZSTD_hintSourceEx(st, C.byref(buf, k * seg), seg, 131072, 3, stable) == 0

        def ahead(c):
            if c % 4 == 0 and (c + 12) * 131072 < len(data):
                assert L.QZSTD_hintSourceEx(st, C.byref(buf, (c + 12) * 131072), seg, 131072, 3, stable) == 0

        got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), 131072, 3, before=ahead)
        assert stats_of(mock, st)[:3] == [24, 0, 6], stats_of(mock, st)
        assert got == oracle_frames(zstd, oracle, data, 131072, 3)
        assert L.QZSTD_hintSourceEx(st, buf, seg, 131072, 3, 2) == -1 and L.QZSTD_hintSourceEx(st, buf, seg, 131072, 3, 0x80000001) == -1
        L.QZSTD_freeSeqProdState(st)
    # the promise matters: a buffer rewritten after a plain announcement is caught by the memcmp (the block is match-found afresh) ...
    other = K.by_name("text", seg, seed=3)
    for stable, caught in ((0, True), (1, False)):
        b2 = (C.c_char * seg).from_buffer_copy(data[:seg])
        st = L.QZSTD_createSeqProdState()
        assert L.QZSTD_hintSourceEx(st, b2, seg, 131072, 3, stable) == 0
        C.memmove(b2, other, seg)
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(b2), seg, 131072, 3)
        served = stats_of(mock, st)[0]
        L.QZSTD_freeSeqProdState(st)
        if caught:
            assert served == 0 and got == oracle_frames(zstd, oracle, other, 131072, 3)
        else:  # ... and with QZSTD_HINT_STABLE it is the caller's responsibility: the stale sequences are served (libzstd may or may not notice)
            assert served > 0


def test_coalescer_many_threads_mixed_levels(mock, zstd, oracle):
    jobs = []
    for t in range(20):
        level = [1, 3, 6][t % 3]
        chunk = [131072, 65536, 100000, 4096][t % 4]
        jobs.append((level, chunk, K.by_name(["text", "binary", "weblog", "mix"][t % 4], chunk * 4 + 13 * t, seed=400 + t)))
    res = [None] * len(jobs)

    def work(t):
        level, chunk, data = jobs[t]
        z = B.Zstd(zstd.path)
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        st = mock.lib.QZSTD_createSeqProdState()
        res[t] = frames_of(z, mock.producer_addr, st, C.addressof(buf), len(data), chunk, level)
        mock.lib.QZSTD_freeSeqProdState(st)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(len(jobs))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for t, (level, chunk, data) in enumerate(jobs):
        assert res[t] == oracle_frames(zstd, oracle, data, chunk, level), "thread %d" % t


def test_one_shot_multi_block_frame_and_stream(mock, zstd):
    data = K.text(9, (1 << 20) + 12345)
    st = mock.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(1, producer=mock.producer_addr, state=st, fallback=False, validate=True)
    frame = zstd.compress2(zc, data)
    zstd.free(zc)
    mock.lib.QZSTD_freeSeqProdState(st)
    assert zstd.decompress(frame, len(data)) == data


def test_modes_through_environment(tmp_path):
    """QZSTD_HIP_COALESCE=0 (a slot per caller, fewer slots than threads), four devices,
    QZSTD_HIP_EXT_REPCODES=1: a fresh process each, frames must round-trip"""
    script = r'''
import sys, ctypes as C, threading
sys.path.insert(0, %r)
import qz_bind as B, qz_corpus as K
plug = B.Plugin(%r)
assert plug.lib.QZSTD_startQatDevice() == 0
z = B.Zstd()
ok = []
datas = [K.by_name("binary" if t & 1 else "weblog", 6 * 131072 + t, seed=t) for t in range(6)]  # generated up front
def work(t):
    data = datas[t]
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    st = plug.lib.QZSTD_createSeqProdState()
    zz = B.Zstd(z.path)
    zc = zz.cctx(3, producer=plug.producer_addr, state=st, fallback=False, validate=True, ext_repcodes=1)
    _, frames = zz.compress_chunks(zc, data, 131072)
    zz.free(zc); plug.lib.QZSTD_freeSeqProdState(st)
    ok.append(b"".join(zz.decompress(f, 131072) for f in frames) == data)
ths = [threading.Thread(target=work, args=(t,)) for t in range(6)]
[t.start() for t in ths]; [t.join() for t in ths]
plug.lib.QZSTD_stopQatDevice()
assert ok == [True] * 6, ok
print("OK")
''' % (os.path.join(ROOT, "tools"), MOCK_SO)
    for env in ({"QZSTD_HIP_COALESCE": "0", "QZSTD_HIP_SLOTS": "3"}, {"QZSTD_MOCK_DEVICES": "4"},
                {"QZSTD_HIP_EXT_REPCODES": "1"}):
        out = subprocess.run(["python", "-c", script], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
        assert out.returncode == 0 and "OK" in out.stdout, (env, out.stderr[-800:])


def test_announced_buffer_that_changes_is_never_served_stale(mock, zstd, oracle):
    """announce two blocks, compress the first, overwrite the buffer, compress the second — the
    callback's bytes no longer equal the staged copy, so the announcement is dropped and the block is match-found
    afresh; and an announcement the caller walks away from does not linger"""
    chunk = 131072
    a, b = K.by_name("text", 2 * chunk, seed=21), K.by_name("weblog", 2 * chunk, seed=22)
    buf = (C.c_char * (2 * chunk)).from_buffer_copy(a)
    st = mock.lib.QZSTD_createSeqProdState()
    assert mock.lib.QZSTD_hintSource(st, buf, 2 * chunk, chunk, 1) == 0

    def overwrite(c):
        if c == 1:
            C.memmove(buf, b, 2 * chunk)

    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), 2 * chunk, chunk, 1, before=overwrite)
    want = oracle_frames(zstd, oracle, a[:chunk] + b[chunk:], chunk, 1)
    assert got == want
    assert zstd.decompress(got[1], chunk) == b[chunk:]
    assert stats_of(mock, st)[:2] == [1, 1]  # block 0 from the announcement, block 1 on its own
    # abandoned announcement: announce 8 blocks, use one, then compress elsewhere — it is dropped after a while
    big = (C.c_char * (8 * chunk)).from_buffer_copy(K.by_name("mix", 8 * chunk, seed=23))
    other = K.by_name("binary", 24 * chunk, seed=24)
    obuf = (C.c_char * len(other)).from_buffer_copy(other)
    assert mock.lib.QZSTD_hintSource(st, big, 8 * chunk, chunk, 1) == 0
    frames_of(zstd, mock.producer_addr, st, C.addressof(big), chunk, chunk, 1)
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(obuf), len(other), chunk, 1)
    assert got == oracle_frames(zstd, oracle, other, chunk, 1)
    mock.lib.QZSTD_freeSeqProdState(st)


def test_time_out_returns_the_error_and_libzstd_falls_back(mock, zstd):
    """reference: 2 s of polling, then ZSTD_SEQUENCE_PRODUCER_ERROR (src/qatseqprod.c:1261-1285) so that
    ZSTD_c_enableSeqProducerFallback takes over; here QZSTD_HIP_TIMEOUT_MS, a stalled (mock) stream, and the stream
    comes back into service once it has drained"""
    import time
    chunk = 65536
    data = K.by_name("text", 4 * chunk, seed=31)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    mock.lib.qzstd_mock_stall_ms.argtypes = [C.c_int]
    with restarted(mock, QZSTD_HIP_TIMEOUT_MS="50"):
        st = mock.lib.QZSTD_createSeqProdState()
        seqs = (B.Sequence * B.sequence_bound(chunk))()
        mock.lib.qzstd_mock_stall_ms(400)
        t0 = time.time()
        rc = mock.lib.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 1, 1 << 17)
        dt = time.time() - t0
        assert rc == B.SEQ_ERROR and 0.04 <= dt < 0.35, (rc, dt)
        # with the fallback enabled libzstd compresses the frame itself while the device is wedged
        zc = zstd.cctx(1, producer=mock.producer_addr, state=st, fallback=True, validate=True)
        frame = zstd.compress2(zc, data[:chunk])
        assert zstd.decompress(frame, chunk) == data[:chunk]
        # an announcement on the wedged device fails the same way; its callbacks then take the per-block path
        mock.lib.qzstd_mock_stall_ms(0)
        time.sleep(0.01)
        rc = mock.lib.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 1, 1 << 17)
        assert rc != B.SEQ_ERROR and rc > 1  # drained: back in service
        zstd.free(zc)
        mock.lib.QZSTD_freeSeqProdState(st)


def test_announcement_is_split_across_the_gpus(mock, zstd, oracle):
    """north star: the batch shards across the GPUs of a node with per-GPU streams and host-side gather (no collective).
    Reference analogue: instances interleaved across devices, src/qatseqprod.c:601-630.  Four mock devices: an
    announcement of 32 blocks becomes four launches, one per device; frames equal the oracle's"""
    chunk = 65536
    data = K.by_name("system", 32 * chunk)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_launches_on.argtypes = [C.c_int]
    with restarted(mock, QZSTD_MOCK_DEVICES="4"):
        before = [L.qzstd_mock_launches_on(d) for d in range(4)]
        st = L.QZSTD_createSeqProdState()
        assert L.QZSTD_hintSource(st, buf, len(data), chunk, 3) == 0
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 3)
        assert stats_of(mock, st)[:2] == [32, 0]
        L.QZSTD_freeSeqProdState(st)
        after = [L.qzstd_mock_launches_on(d) for d in range(4)]
        assert [a - b for a, b in zip(after, before)] == [1, 1, 1, 1]
        # QZSTD_HIP_SPLIT=1 keeps an announcement on the state's own GPU
    with restarted(mock, QZSTD_MOCK_DEVICES="4", QZSTD_HIP_SPLIT="1"):
        before = sum(L.qzstd_mock_launches_on(d) for d in range(4))
        st = L.QZSTD_createSeqProdState()
        assert L.QZSTD_hintSource(st, buf, len(data), chunk, 3) == 0
        got1 = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 3)
        L.QZSTD_freeSeqProdState(st)
        assert sum(L.qzstd_mock_launches_on(d) for d in range(4)) - before == 1
    want = oracle_frames(zstd, oracle, data, chunk, 3)
    assert got == want and got1 == want


def test_numa_states_pick_a_gpu_of_their_socket_and_buffers_follow_the_gpu(mock, zstd, oracle):
    """four mock GPUs on two fake NUMA nodes (QZSTD_MOCK_NODES=2: GPU d on node d % 2; reference: qaeMemAllocNUMA per instance,
    src/qatseqprod.c:216-246).  Threads that run on node 1 (QZSTD_HIP_NUMA_NODE=1 says so; the kernel is asked otherwise) get GPUs 1 and 3
    only, round-robin; every pinned buffer of a GPU is allocated on that GPU's node; QZSTD_HIP_NUMA=0 gives plain round-robin over all
    four and unplaced memory; QZSTD_deviceStats reports the node map; frames stay the oracle's"""
    chunk = 65536
    data = K.by_name("system", 6 * chunk, seed=5)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_node_allocs.restype = C.c_ulong
    L.qzstd_mock_node_allocs.argtypes = [C.c_int]
    L.QZSTD_deviceStats.argtypes = [C.c_int, C.POINTER(C.c_ulong * 4)]
    want = oracle_frames(zstd, oracle, data, chunk, 1)

    def run_states(n):
        used = []
        for _ in range(n):
            before = [dev_stats(d)[2] + dev_stats(d)[1] for d in range(4)]
            st = L.QZSTD_createSeqProdState()
            assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == want
            L.QZSTD_freeSeqProdState(st)
            after = [dev_stats(d)[2] + dev_stats(d)[1] for d in range(4)]
            used.append([d for d in range(4) if after[d] != before[d]])
        return used

    def dev_stats(d):
        ds = (C.c_ulong * 4)()
        assert L.QZSTD_deviceStats(d, C.byref(ds)) == 4
        return list(ds)

    with restarted(mock, QZSTD_MOCK_DEVICES="4", QZSTD_MOCK_NODES="2", QZSTD_HIP_NUMA_NODE="1"):
        L.qzstd_mock_node_allocs_reset()
        assert [dev_stats(d)[3] for d in range(4)] == [1, 2, 1, 2]  # node + 1
        used = run_states(4)
        assert used == [[1], [3], [1], [3]], used                # GPUs of node 1 only, round-robin
        assert L.qzstd_mock_node_allocs(1) > 0 and L.qzstd_mock_node_allocs(0) == 0 and L.qzstd_mock_node_allocs(-1) == 0
    with restarted(mock, QZSTD_MOCK_DEVICES="4", QZSTD_MOCK_NODES="2", QZSTD_HIP_NUMA_NODE="0"):
        assert run_states(2) == [[0], [2]]
    with restarted(mock, QZSTD_MOCK_DEVICES="4", QZSTD_MOCK_NODES="2", QZSTD_HIP_NUMA_NODE="1", QZSTD_HIP_NUMA="0"):
        L.qzstd_mock_node_allocs_reset()
        assert run_states(4) == [[0], [1], [2], [3]]             # plain round-robin
        assert L.qzstd_mock_node_allocs(0) == 0 and L.qzstd_mock_node_allocs(1) == 0 and L.qzstd_mock_node_allocs(-1) > 0
    with restarted(mock, QZSTD_MOCK_DEVICES="4", QZSTD_MOCK_NODES="2", QZSTD_HIP_NUMA_NODE="5"):
        assert run_states(3) == [[0], [1], [2]]                   # a node without a GPU: round-robin over all of them
    # an announcement's pinned buffers go next to the state's own GPU; the split still reaches every GPU
    with restarted(mock, QZSTD_MOCK_DEVICES="4", QZSTD_MOCK_NODES="2", QZSTD_HIP_NUMA_NODE="1"):
        L.qzstd_mock_node_allocs_reset()
        big = K.by_name("system", 32 * chunk)
        bbuf = (C.c_char * len(big)).from_buffer_copy(big)
        st = L.QZSTD_createSeqProdState()
        assert L.QZSTD_hintSource(st, bbuf, len(big), chunk, 1) == 0
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(bbuf), len(big), chunk, 1)
        L.QZSTD_freeSeqProdState(st)
        assert got == oracle_frames(zstd, oracle, big, chunk, 1)
        assert L.qzstd_mock_node_allocs(1) >= 4 and L.qzstd_mock_node_allocs(0) == 0
        assert all(dev_stats(d)[0] == 8 for d in range(4))        # 32 blocks announced, 8 per GPU


def test_dense_block_is_redone_alone(mock, zstd, oracle):
    """a block with more sequences than a batch's result pitch (16384) is redone with the caller's full capacity"""
    import random
    rng = random.Random(7)
    words = [bytes(rng.randrange(97, 123) for _ in range(5)) for _ in range(200)]
    data = b"".join(rng.choice(words) + bytes([rng.randrange(256)]) for _ in range(131072 // 6 + 1))[:131072]
    n, _ = oracle.find(oracle.profile(1, 131072), data, cap=B.sequence_bound(131072))
    assert n > 16384, n
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    st = mock.lib.QZSTD_createSeqProdState()
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), 131072, 1)
    mock.lib.QZSTD_freeSeqProdState(st)
    assert got == oracle_frames(zstd, oracle, data, 131072, 1)


def test_dense_block_does_not_end_its_announcement(mock, zstd, oracle):
    """round-5 ADVICE: a STABLE announcement used to be dropped WHOLE at the first block it could not serve — one dense block (more
    sequences than the result pitch) in a claim and the rest of the claim went through the per-block path, after a synchronous wait for
    every launch in flight.  Now only that block takes the per-block path; the announcement serves the blocks behind it.  A verified
    announcement over addresses a newer one names keeps serving by CONTENT (streaming callers refill and re-announce one buffer)."""
    import random
    rng = random.Random(7)
    words = [bytes(rng.randrange(97, 123) for _ in range(5)) for _ in range(200)]
    dense = b"".join(rng.choice(words) + bytes([rng.randrange(256)]) for _ in range(131072 // 6 + 1))[:131072]
    assert oracle.find(oracle.profile(1, 131072), dense, cap=B.sequence_bound(131072))[0] > 16384
    plain = K.by_name("system", 7 * 131072)
    data = plain[:2 * 131072] + dense + plain[2 * 131072:]  # 8 blocks, the third one dense
    want = oracle_frames(zstd, oracle, data, 131072, 1)
    for stable in (1, 0):
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        st = mock.lib.QZSTD_createSeqProdState()
        assert mock.lib.QZSTD_hintSourceEx(st, buf, len(data), 131072, 1, stable) == 0
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), 131072, 1)
        stats = stats_of(mock, st)
        mock.lib.QZSTD_freeSeqProdState(st)
        assert got == want
        assert stats[0] == 7 and stats[1] == 1, (stable, stats)  # seven blocks from the announcement, the dense one alone
    # a verified announcement whose addresses are announced again: by content from then on, not dropped
    data2 = K.by_name("text", 4 * 131072, seed=11)
    buf = (C.c_char * len(data2)).from_buffer_copy(data2)
    other = (C.c_char * len(data2)).from_buffer_copy(data2)  # the same bytes elsewhere (libzstd's window, for a streaming caller)
    st = mock.lib.QZSTD_createSeqProdState()
    assert mock.lib.QZSTD_hintSource(st, buf, len(data2), 131072, 1) == 0
    assert mock.lib.QZSTD_hintSource(st, C.byref(buf, 131072), 2 * 131072, 131072, 1) == 0  # overlaps the first
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(other), len(data2), 131072, 1)
    stats = stats_of(mock, st)
    mock.lib.QZSTD_freeSeqProdState(st)
    assert got == oracle_frames(zstd, oracle, data2, 131072, 1)
    assert stats[0] == 4 and stats[1] == 0, stats  # all four by content: the older announcement was still there for blocks 0 and 3


def test_batch_front_end_over_the_mock(mock, zstd, oracle):
    """include/qzstd_frontend.h: a pool of CCtx threads fed from one chunk cursor, two claims kept
    announced ahead; frames are the oracle's, every block comes from an announcement"""
    front_so = os.path.join(ROOT, "tests", "mock", "libqzstdfront_mock.so")
    build_shared(["gcc", "-O2", "-g", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-DQZ_TEST_HOOKS", "-shared", "-fPIC", "-pthread",
                  "-I" + os.path.join(ROOT, "include"), "-o", front_so,
                           os.path.join(B.PKG_DIR, "frontend", "qzstd_frontend.c"), MOCK_SO, zstd.path,
                           "-Wl,-rpath," + os.path.dirname(MOCK_SO), "-Wl,-rpath," + os.path.dirname(zstd.path)], front_so)
    F = C.CDLL(front_so)

    class Params(C.Structure):
        _fields_ = [("nThreads", C.c_int), ("level", C.c_int), ("chunkSize", C.c_size_t), ("segmentBytes", C.c_size_t),
                    ("extRepcodes", C.c_int), ("useProducer", C.c_int)]

    F.QZSTD_createFront.restype = C.c_void_p
    F.QZSTD_createFront.argtypes = [C.POINTER(Params)]
    F.QZSTD_frontFrameStride.restype = C.c_size_t
    F.QZSTD_frontFrameStride.argtypes = [C.c_void_p]
    F.QZSTD_frontCompress.restype = C.c_size_t
    F.QZSTD_frontCompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    F.QZSTD_frontCompact.restype = C.c_size_t
    F.QZSTD_frontCompact.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t]
    F.QZSTD_frontStats.argtypes = [C.c_void_p, C.POINTER(C.c_ulong)]
    F.QZSTD_freeFront.argtypes = [C.c_void_p]
    chunk = 65536
    data = K.by_name("system", 37 * chunk + 1234)
    for level, threads in ((1, 3), (6, 5)):
        prm = Params(threads, level, chunk, 8 * chunk, 0, 1)
        f = F.QZSTD_createFront(C.byref(prm))
        assert f
        stride = F.QZSTD_frontFrameStride(f)
        n = (len(data) + chunk - 1) // chunk
        dst = C.create_string_buffer(n * stride)
        sizes = (C.c_size_t * n)()
        for _ in range(2):  # the pool is persistent: a second job on the same front
            assert F.QZSTD_frontCompress(f, data, len(data), dst, len(dst), sizes) == n
        frames = [dst.raw[c * stride:c * stride + sizes[c]] for c in range(n)]
        assert frames == oracle_frames(zstd, oracle, data, chunk, level)
        st = (C.c_ulong * 2)()
        F.QZSTD_frontStats(f, st)
        assert st[0] == 2 * n and st[1] == 0, list(st)
        total = F.QZSTD_frontCompact(f, dst, sizes, n)
        assert total == sum(sizes) and dst.raw[:total] == b"".join(frames)
        F.QZSTD_freeFront(f)
    assert F.QZSTD_createFront(C.byref(Params(0, 1, chunk, 0, 0, 1))) is None


def _front_lib(zstd):
    front_so = os.path.join(ROOT, "tests", "mock", "libqzstdfront_mock.so")
    build_shared(["gcc", "-O2", "-g", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-DQZ_TEST_HOOKS", "-shared", "-fPIC", "-pthread",
                  "-I" + os.path.join(ROOT, "include"), "-o", front_so,
                           os.path.join(B.PKG_DIR, "frontend", "qzstd_frontend.c"), MOCK_SO, zstd.path,
                           "-Wl,-rpath," + os.path.dirname(MOCK_SO), "-Wl,-rpath," + os.path.dirname(zstd.path)], front_so)
    F = C.CDLL(front_so)
    F.QZSTD_createFront.restype = C.c_void_p
    F.QZSTD_createFront.argtypes = [C.POINTER(B.FrontParams)]
    F.QZSTD_frontFrameStride.restype = C.c_size_t
    F.QZSTD_frontFrameStride.argtypes = [C.c_void_p]
    F.QZSTD_frontCompress.restype = C.c_size_t
    F.QZSTD_frontCompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    F.QZSTD_freeFront.argtypes = [C.c_void_p]
    return F


@pytest.mark.parametrize("tail", [3, 1234, 0])
@pytest.mark.parametrize("threads", [1, 3])
def test_front_end_jobs_over_a_reused_buffer(mock, zstd, oracle, tail, threads):
    """round-4 ADVICE (high): the front-end announces with QZSTD_HINT_STABLE, and an announcement used to live until the callback
    of its last block — which never comes for a last chunk below 7 bytes (libzstd does not ask the producer).  The stale announcement
    then served the NEXT job by address when the caller used the same buffer again: frames that did not decode to the input.
    Now every worker ends its announcements with its job (QZSTD_dropHints)."""
    F = _front_lib(zstd)
    chunk = 65536
    size = 3 * chunk + tail
    a = K.by_name("system", size, seed=41)
    b = K.by_name("text", size, seed=42)
    buf = (C.c_char * size).from_buffer_copy(a)
    prm = B.FrontParams(threads, 1, chunk, 4 * chunk, 0, 1)
    f = F.QZSTD_createFront(C.byref(prm))
    assert f
    stride = F.QZSTD_frontFrameStride(f)
    n = (size + chunk - 1) // chunk
    dst = C.create_string_buffer(n * stride)
    sizes = (C.c_size_t * n)()
    try:
        for content in (a, b, a, b):
            C.memmove(buf, content, size)
            assert F.QZSTD_frontCompress(f, buf, size, dst, len(dst), sizes) == n
            frames = [dst.raw[c * stride:c * stride + sizes[c]] for c in range(n)]
            back = b"".join(zstd.decompress(fr, chunk) for fr in frames)
            assert back == content
            assert frames == oracle_frames(zstd, oracle, content, chunk, 1)
    finally:
        F.QZSTD_freeFront(f)


def test_stable_announcement_has_a_bounded_life(mock, zstd, oracle):
    """round-4 ADVICE (medium): a STABLE announcement serves every block once, going forward; it ends when a block is asked for a
    second time, when a newer announcement names its addresses, at QZSTD_dropHints; the newest announcement is looked at first; the
    sampled comparison (every 16th block) reveals a broken promise (QZSTD_hintBroken)"""
    L = mock.lib
    L.QZSTD_hintSourceEx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]
    L.QZSTD_dropHints.argtypes = [C.c_void_p]
    L.QZSTD_hintBroken.argtypes = [C.c_void_p]
    L.QZSTD_hintBroken.restype = C.c_ulong
    chunk, nb = 65536, 8
    a = K.by_name("system", nb * chunk, seed=51)
    b = K.by_name("text", nb * chunk, seed=52)
    buf = (C.c_char * len(a)).from_buffer_copy(a)
    addr = C.addressof(buf)
    # 1. a block asked for again (the buffer is being used again, the last blocks' callbacks never came): dropped, fresh match-finding
    st = L.QZSTD_createSeqProdState()
    assert L.QZSTD_hintSourceEx(st, buf, len(a), chunk, 1, 1) == 0
    got = frames_of(zstd, mock.producer_addr, st, addr, 5 * chunk, chunk, 1)  # walks 5 of the 8 blocks, then away
    assert stats_of(mock, st)[0] == 5
    C.memmove(buf, b, len(b))
    got = frames_of(zstd, mock.producer_addr, st, addr, len(b), chunk, 1)      # a new job over the same memory, nothing announced
    assert got == oracle_frames(zstd, oracle, b, chunk, 1)
    assert stats_of(mock, st)[0] == 5  # nothing more was served from the stale announcement
    L.QZSTD_freeSeqProdState(st)
    # 2. a newer announcement over the same addresses voids the older one, and the newest is looked at first
    C.memmove(buf, a, len(a))
    st = L.QZSTD_createSeqProdState()
    assert L.QZSTD_hintSourceEx(st, buf, len(a), chunk, 1, 1) == 0
    C.memmove(buf, b, len(b))
    assert L.QZSTD_hintSourceEx(st, buf, len(b), chunk, 1, 1) == 0
    got = frames_of(zstd, mock.producer_addr, st, addr, len(b), chunk, 1)
    assert got == oracle_frames(zstd, oracle, b, chunk, 1) and stats_of(mock, st)[0] == nb
    L.QZSTD_freeSeqProdState(st)
    # 3. QZSTD_dropHints ends everything announced
    C.memmove(buf, a, len(a))
    st = L.QZSTD_createSeqProdState()
    assert L.QZSTD_hintSourceEx(st, buf, len(a), chunk, 1, 1) == 0
    L.QZSTD_dropHints(st)
    C.memmove(buf, b, len(b))
    got = frames_of(zstd, mock.producer_addr, st, addr, len(b), chunk, 1)
    assert got == oracle_frames(zstd, oracle, b, chunk, 1) and stats_of(mock, st)[0] == 0
    L.QZSTD_dropHints(None)
    L.QZSTD_freeSeqProdState(st)
    # 4. the sampled check: 40 blocks announced STABLE and then rewritten — by the 16th block served the library notices, drops the
    # announcement, counts it, and everything from there on is match-found afresh
    big_a = K.by_name("system", 40 * chunk, seed=53)
    big_b = K.by_name("mix", 40 * chunk, seed=54)
    bb = (C.c_char * len(big_a)).from_buffer_copy(big_a)
    st = L.QZSTD_createSeqProdState()
    for k in range(0, 40, 10):
        assert L.QZSTD_hintSourceEx(st, C.byref(bb, k * chunk), 10 * chunk, chunk, 1, 1) == 0
    C.memmove(bb, big_b, len(big_b))
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(bb), len(big_b), chunk, 1)
    want = oracle_frames(zstd, oracle, big_b, chunk, 1)
    assert L.QZSTD_hintBroken(st) >= 1
    assert got[16:20] == want[16:20]      # the rest of the announcement that was caught: afresh
    assert stats_of(mock, st)[0] < 40
    L.QZSTD_freeSeqProdState(st)
    # ... and a caller that keeps the promise is never counted
    st = L.QZSTD_createSeqProdState()
    for k in range(0, 40, 10):
        assert L.QZSTD_hintSourceEx(st, C.byref(bb, k * chunk), 10 * chunk, chunk, 1, 1) == 0
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(bb), len(big_b), chunk, 1)
    assert got == want and L.QZSTD_hintBroken(st) == 0 and stats_of(mock, st)[0] == 40
    L.QZSTD_freeSeqProdState(st)


def test_streaming_caller_is_served_from_an_announcement_by_content(mock, zstd, oracle):
    """ZSTD_compressStream2 with small feeds: libzstd hands the producer blocks out of its own window buffer, so the
    callback's address never lies inside what the caller announced.  An announced grid block with the same bytes serves
    it all the same (fingerprint + memcmp): every full block of the stream comes from the announcement."""
    L = zstd.lib

    class InB(C.Structure):
        _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    class OutB(C.Structure):
        _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(OutB), C.POINTER(InB), C.c_int]
    L.ZSTD_compressStream2.restype = C.c_size_t
    data = K.by_name("system", 9 * 131072 + 5000)
    announced = (C.c_char * len(data)).from_buffer_copy(data)  # what the caller read the file into ...
    feed = (C.c_char * len(data)).from_buffer_copy(data)       # ... and a different buffer it feeds libzstd from
    st = mock.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(1, producer=mock.producer_addr, state=st, fallback=False, validate=True, blockSplitterLevel=1)
    assert mock.lib.QZSTD_hintSource(st, announced, len(data), 131072, 1) == 0
    dst = C.create_string_buffer(L.ZSTD_compressBound(len(data)))
    out = OutB(C.addressof(dst), len(dst), 0)
    pos = 0
    while pos < len(data):
        n = min(50000, len(data) - pos)
        inb = InB(C.addressof(feed) + pos, n, 0)
        while inb.pos < inb.size:
            r = L.ZSTD_compressStream2(zc, C.byref(out), C.byref(inb), B.e_continue)
            assert not zstd.is_error(r), zstd.err(r)
        pos += n
    inb = InB(None, 0, 0)
    while True:
        r = L.ZSTD_compressStream2(zc, C.byref(out), C.byref(inb), B.e_end)
        assert not zstd.is_error(r), zstd.err(r)
        if r == 0:
            break
    stats = stats_of(mock, st)
    zstd.free(zc)
    mock.lib.QZSTD_freeSeqProdState(st)
    assert zstd.decompress(dst.raw[:out.pos], len(data)) == data
    assert stats[0] >= 9 and stats[0] + stats[1] == 10, stats  # the nine full blocks (and maybe the tail) by content


def test_replay_ceiling_tool_over_the_mock(mock, zstd, tmp_path):
    """test/replaybench.c (the ceiling of any external producer: recorded sequences replayed by memcpy): over the mock every
    block is recorded from the real producer path, replayed from several threads, and the frames round-trip — same size as
    the producer path itself gives."""
    exe = str(tmp_path / "replaybench")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(B.PKG_DIR, "test", "replaybench.c"), MOCK_SO, zstd.path,
                           "-Wl,-rpath," + os.path.dirname(MOCK_SO), "-Wl,-rpath," + os.path.dirname(zstd.path), "-lpthread"])
    f = tmp_path / "in.bin"
    data = K.by_name("system", 20 * 131072 + 777, seed=2)
    f.write_bytes(data)
    for args, nrec in ((["-t3", "-l1", "-c128K", "-L1"], 21), (["-t2", "-l1", "-c256K", "-L6"], 21), (["-t2", "-l1", "-c32K", "-L12"], 81)):
        out = subprocess.run([exe] + args + [str(f)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "round trip PASS" in out.stdout, out.stdout + out.stderr
        assert "%d of %d blocks recorded" % (nrec, nrec) in out.stdout, out.stdout


# ---------------------------------------------------------------- round 3: the resident service, fallback counters, parked buffers
@pytest.fixture(autouse=True)
def _service_in_service(mock):
    """a test that lets a service request time out leaves the (mock) service marked broken: every test starts with it repaired"""
    mock.lib.qzstd_mock_service_repair()
    yield


def fail_stats(plug, st):
    s = (C.c_ulong * 8)()
    plug.lib.QZSTD_failStats(st, C.byref(s))
    return list(s)


@pytest.mark.parametrize("level,chunk", [(1, 131072), (2, 65536), (1, 100001), (1, 4096), (2, 1000)])
def test_service_path_serves_unchanged_callers(mock, zstd, oracle, level, chunk):
    """per-block requests of the levels the resident service serves go through qzstd_hip_service_submit (mocked: the oracle
    per work item, 4 KiB items, counts written last) and the joined lists give the oracle's frames; with QZSTD_HIP_SERVICE=0
    the same callers take the launch path — same frames"""
    data = K.by_name("mix", 9 * chunk + 321, seed=level + 40)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    want = oracle_frames(zstd, oracle, data, chunk, level)
    before = L.qzstd_mock_service_requests()
    st = L.QZSTD_createSeqProdState()
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, level)
    fs, hs = fail_stats(mock, st), stats_of(mock, st)
    L.QZSTD_freeSeqProdState(st)
    assert got == want
    assert L.qzstd_mock_service_requests() - before == 10 and fs[7] == 10 and hs[1] == 10 and fs[0] == 0, (fs, hs)
    with restarted(mock, QZSTD_HIP_SERVICE="0"):
        before = L.qzstd_mock_service_requests()
        st = L.QZSTD_createSeqProdState()
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, level)
        fs = fail_stats(mock, st)
        L.QZSTD_freeSeqProdState(st)
        assert got == want and L.qzstd_mock_service_requests() == before and fs[7] == 0


def test_service_entries_that_arrive_after_their_count(mock, zstd, oracle):
    """the count says how many entries an item has, not that they are there: the mock stores the counts first and marks the entries (the
    request's epoch in their fourth word) up to a couple of milliseconds later, last entry first — the join waits for every entry's mark
    and the frames are the oracle's"""
    chunk = 131072
    data = K.by_name("system", 3 * chunk + 777, seed=8)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_late_marks.argtypes = [C.c_int]
    L.qzstd_mock_late_marks(1)
    try:
        st = L.QZSTD_createSeqProdState()
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1)
        fs = fail_stats(mock, st)
        L.QZSTD_freeSeqProdState(st)
    finally:
        L.qzstd_mock_late_marks(0)
    assert got == oracle_frames(zstd, oracle, data, chunk, 1)
    assert fs[7] == 4 and fs[0] == 0, fs


def test_announced_entries_that_arrive_after_their_count(mock, zstd, oracle):
    """round 4: an announcement's launch is complete when its blocks' COUNT WORDS are in (no stream query) — and a count says how many entries a
    block has, not that they have arrived: the mock publishes the counts at once and the entries up to half a millisecond later, last entry
    first, garbage in their place until then.  Every entry carries the announcement's epoch in its fourth word and is taken when it shows it:
    frames are the oracle's, every block came from the announcement, nothing failed — for plain, stable and four-deep announcements, over
    buffers that are reused (the epoch of the previous announcement in the same place does not count)"""
    chunk = 65536
    data = K.by_name("system", 24 * chunk + 333, seed=12)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_late_marks.argtypes = [C.c_int]
    L.QZSTD_hintSourceEx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]
    seg = 4 * chunk
    want = oracle_frames(zstd, oracle, data, chunk, 3)
    L.qzstd_mock_late_marks(1)
    try:
        for stable in (0, 1):
            st = L.QZSTD_createSeqProdState()
            for rnd in range(2):  # the second round reuses the four announcements' result areas
                for k in range(3):
                    assert L.QZSTD_hintSourceEx(st, C.byref(buf, k * seg), seg, chunk, 3, stable) == 0

                def ahead(c):
                    o = (c + 12) * chunk
                    if c % 4 == 0 and o < len(data):
                        assert L.QZSTD_hintSourceEx(st, C.byref(buf, o), min(seg, len(data) - o), chunk, 3, stable) == 0

                got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 3, before=ahead)
                assert got == want, "round %d, stable %d" % (rnd, stable)
            served = stats_of(mock, st)
            fs = fail_stats(mock, st)
            L.QZSTD_freeSeqProdState(st)
            assert served[0] == 2 * 25 and served[1] == 0 and fs[0] == 0, (served, fs)
    finally:
        L.qzstd_mock_late_marks(0)


def test_announcement_epoch_wrap_wipes_stale_marks(mock, zstd, oracle):
    """the mark of an announcement's entries is a 24-bit epoch per announcement buffer: when it starts over the result area is wiped once (an
    entry nothing overwrote for a lap would show a valid mark again); announcements across the wrap serve the oracle's frames"""
    chunk = 65536
    data = K.by_name("mix", 16 * chunk, seed=14)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_test_set_hint_epochs.argtypes = [C.c_void_p, C.c_uint]
    want = oracle_frames(zstd, oracle, data, chunk, 1)
    st = L.QZSTD_createSeqProdState()
    assert L.QZSTD_hintSource(st, buf, len(data), chunk, 1) == 0  # (buffers exist before the epochs are moved)
    assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == want
    L.qzstd_test_set_hint_epochs(st, 0xFFFFFE)
    for _ in range(6):  # epochs 0xFFFFFF, then the wrap to 1, on each of the four buffers in turn
        assert L.QZSTD_hintSource(st, buf, len(data), chunk, 1) == 0
        assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == want
    served, fs = stats_of(mock, st), fail_stats(mock, st)
    L.QZSTD_freeSeqProdState(st)
    assert served[0] == 7 * 16 and served[1] == 0 and fs[0] == 0, (served, fs)


def test_packed_and_sixteen_byte_announcement_entries(mock, zstd, oracle):
    """round 6: announcements ask for PACKED result entries by default (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT: 8 bytes with a 12-bit tag, half the
    bytes the kernel writes over PCIe); QZSTD_HIP_HINT_COMPACT=0 keeps the 16-byte entries with the 24-bit epoch.  Both forms, with entries
    that arrive after their counts (the mock's late-marks hook), over reused buffers and across the tags' lap (the result area is wiped when the
    12-bit tag starts over: 4 095 announcements per buffer): frames are the oracle's, every block comes from an announcement"""
    chunk = 65536
    data = K.by_name("system", 12 * chunk + 777, seed=31)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_late_marks.argtypes = [C.c_int]
    L.qzstd_test_set_hint_epochs.argtypes = [C.c_void_p, C.c_uint]
    want = oracle_frames(zstd, oracle, data, chunk, 1)
    for compact in ("1", "0"):
        with restarted(mock, QZSTD_HIP_HINT_COMPACT=compact):
            st = L.QZSTD_createSeqProdState()
            assert L.QZSTD_hintSource(st, buf, len(data), chunk, 1) == 0
            assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == want
            L.qzstd_test_set_hint_epochs(st, 0xFFC if compact == "1" else 0xFFFFFC)  # a few announcements before the lap
            L.qzstd_mock_late_marks(1)
            try:
                for _ in range(8):
                    assert L.QZSTD_hintSource(st, buf, len(data), chunk, 1) == 0
                    assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == want, compact
            finally:
                L.qzstd_mock_late_marks(0)
            served, fs = stats_of(mock, st), fail_stats(mock, st)
            L.QZSTD_freeSeqProdState(st)
            assert served[0] == 9 * 13 and served[1] == 0 and fs[0] == 0, (compact, served, fs)


def test_service_epoch_wrap_wipes_stale_marks(mock, zstd, oracle):
    """the request epoch is 24 bits per slot: when it starts over, entries no request of the last lap overwrote would carry a mark
    that is valid again — the slot's result area is wiped once per lap (round-3 verdict, weak 3).  Every slot is put just before the
    wrap (test hook of the mock build), blocks are served across it: frames are the oracle's, nothing fails"""
    chunk = 131072
    data = K.by_name("system", 6 * chunk + 99, seed=14)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_test_set_service_epochs.argtypes = [C.c_uint]
    st = L.QZSTD_createSeqProdState()
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), 2 * chunk, chunk, 1)  # (slots exist, results written once)
    L.qzstd_test_set_service_epochs(0xFFFFFD)
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1)
    fs = fail_stats(mock, st)
    L.QZSTD_freeSeqProdState(st)
    assert got == oracle_frames(zstd, oracle, data, chunk, 1)
    assert fs[0] == 0 and fs[7] >= 7, fs


def test_service_coarser_items_and_other_levels(mock, zstd, oracle):
    """QZSTD_HIP_SERVICE_ITEM: items of several segments; levels 3-4 and the chain levels (6, 12: a scratch per item) are served
    too; requests the dispatcher hands back (another level is resident) take the launch path"""
    chunk = 131072
    data = K.by_name("system", 3 * chunk, seed=3)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_service_level.argtypes = [C.c_int]
    with restarted(mock, QZSTD_HIP_SERVICE_ITEM="16384"):
        st = L.QZSTD_createSeqProdState()
        assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == oracle_frames(zstd, oracle, data, chunk, 1)
        assert fail_stats(mock, st)[7] == 3
        L.QZSTD_freeSeqProdState(st)
    for level, served in ((3, 1), (6, 1), (12, 1)):
        st = L.QZSTD_createSeqProdState()
        assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), chunk, chunk, level) == oracle_frames(zstd, oracle, data[:chunk], chunk, level)
        assert fail_stats(mock, st)[7] == served and stats_of(mock, st)[1] == 1
        L.QZSTD_freeSeqProdState(st)
    with restarted(mock, QZSTD_HIP_SERVICE="0"):  # the batches' own items: eight per block
        st = L.QZSTD_createSeqProdState()
        assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 6) == oracle_frames(zstd, oracle, data, chunk, 6)
        assert fail_stats(mock, st)[7] == 0
        L.QZSTD_freeSeqProdState(st)
    L.qzstd_mock_service_level(2)  # "level 2 is resident": level-1 requests come back rejected
    try:
        st = L.QZSTD_createSeqProdState()
        assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1) == oracle_frames(zstd, oracle, data, chunk, 1)
        fs = fail_stats(mock, st)
        assert fs[7] == 0 and fs[0] == 0 and stats_of(mock, st)[1] == 3, fs
        L.QZSTD_freeSeqProdState(st)
    finally:
        L.qzstd_mock_service_level(0)


@pytest.mark.parametrize("level,chunk", [(1, 131072), (2, 100001), (6, 131072), (12, 32768), (1, 5000)])
def test_service_progressive_staging(mock, zstd, oracle, level, chunk):
    """round 5 (include/qzstd_hip.h: QZSTD_HIP_NSEQ_STAGING): where the device layer's workers look at a count word before they read
    its slice, the host queues the request FIRST and copies the block into the pinned buffer behind it, slice by slice.  The mock
    with QZSTD_MOCK_PROGRESSIVE=1 serves requests on a thread of its own and waits for every slice like the resident kernels do:
    frames are the oracle's, every block went through the service; a request the dispatcher hands back while the host is still
    staging (another level is resident) takes the launch path"""
    data = K.by_name("system", 5 * chunk + 77, seed=level)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_service_level.argtypes = [C.c_int]
    nblk = (len(data) + chunk - 1) // chunk
    with restarted(mock, QZSTD_MOCK_PROGRESSIVE="1"):
        assert L.qzstd_hip_service_progressive(0) == 1
        st = L.QZSTD_createSeqProdState()
        for _ in range(2):
            got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, level)
            assert got == oracle_frames(zstd, oracle, data, chunk, level)
        fs = fail_stats(mock, st)
        assert fs[0] == 0 and fs[7] == 2 * nblk, fs
        L.QZSTD_freeSeqProdState(st)
        L.qzstd_mock_service_level(level + 1 if level < 12 else 1)  # handed back: the words say REJECTED, not STAGING, when the host gets there
        try:
            st = L.QZSTD_createSeqProdState()
            assert frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, level) == oracle_frames(zstd, oracle, data, chunk, level)
            fs = fail_stats(mock, st)
            assert fs[7] == 0 and fs[0] == 0 and stats_of(mock, st)[1] == nblk, fs
            L.QZSTD_freeSeqProdState(st)
        finally:
            L.qzstd_mock_service_level(0)
    assert L.qzstd_hip_service_progressive(0) == 0


def test_service_time_out_is_an_error_then_the_launch_path_takes_over(mock, zstd):
    """a service request whose counts never arrive: the error code after QZSTD_HIP_TIMEOUT_MS (reference: 2 s of polling,
    src/qatseqprod.c:1261-1285), counted as a time-out; the service is not used again, the slot not before its counts have
    arrived; later blocks are served by the launch path"""
    import time
    chunk = 65536
    data = K.by_name("text", 2 * chunk, seed=33)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_stall_ms.argtypes = [C.c_int]
    with restarted(mock, QZSTD_HIP_TIMEOUT_MS="40", QZSTD_HIP_COALESCE="1"):
        st = L.QZSTD_createSeqProdState()
        seqs = (B.Sequence * B.sequence_bound(chunk))()
        L.qzstd_mock_stall_ms(200)
        t0 = time.time()
        rc = L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 1, 1 << 17)
        assert rc == B.SEQ_ERROR and 0.03 <= time.time() - t0 < 0.19
        fs = fail_stats(mock, st)
        assert fs[0] == 1 and fs[3] == 1, fs
        L.qzstd_mock_stall_ms(0)
        time.sleep(0.25)
        rc = L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 1, 1 << 17)
        assert rc != B.SEQ_ERROR and rc > 1
        fs = fail_stats(mock, st)
        assert fs[0] == 1 and fs[7] == 0 and stats_of(mock, st)[1] == 1, fs  # served, by the launch path
        L.QZSTD_freeSeqProdState(st)
    L.qzstd_mock_service_repair()


def test_fail_stats_count_every_cause(mock, zstd):
    """QZSTD_failStats: guards, capacity rule, time-outs — what libzstd's fallback would otherwise hide"""
    chunk = 32768
    data = K.by_name("text", chunk, seed=35)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    st = L.QZSTD_createSeqProdState()
    seqs = (B.Sequence * B.sequence_bound(chunk))()
    assert L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 13, 1 << 17) == B.SEQ_ERROR   # level
    assert L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, buf, 8, 1, 1 << 17) == B.SEQ_ERROR     # dictionary
    assert L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 1, 1024) == B.SEQ_ERROR       # window
    assert L.qatSequenceProducer(st, seqs, 64, buf, chunk, None, 0, 1, 1 << 17) == B.SEQ_ERROR           # 64 entries cannot hold it
    assert L.qatSequenceProducer(st, seqs, 64, buf, chunk, None, 0, 6, 1 << 17) == B.SEQ_ERROR           # ... on the launch path neither
    assert L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 1, 1 << 17) > 1
    fs = fail_stats(mock, st)
    assert fs[0] == 5 and fs[1] == 3 and fs[4] == 2 and fs[2] == fs[3] == fs[5] == 0, fs
    L.QZSTD_freeSeqProdState(st)


def test_timed_out_announcement_parks_its_buffers(mock, zstd, oracle):
    """round-2 ADVICE: a part of an announcement that timed out may still be read and written by its kernel — the announcement's
    pinned buffers are parked (neither reused, scrubbed nor freed) until the stream has drained, the next announcement gets
    fresh ones, and its frames are the oracle's"""
    import time
    chunk = 65536
    data = K.by_name("system", 8 * chunk, seed=5)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    L = mock.lib
    L.qzstd_mock_stall_ms.argtypes = [C.c_int]
    L.qzstd_test_orphans.restype = C.c_ulong
    with restarted(mock, QZSTD_HIP_TIMEOUT_MS="30"):
        before = L.qzstd_test_orphans()
        st = L.QZSTD_createSeqProdState()
        L.qzstd_mock_stall_ms(300)  # (before the announcement: a stalled launch never publishes its blocks' count words)
        assert L.QZSTD_hintSource(st, buf, len(data), chunk, 3) == 0
        seqs = (B.Sequence * B.sequence_bound(chunk))()
        assert L.qatSequenceProducer(st, seqs, len(seqs), buf, chunk, None, 0, 3, 1 << 17) == B.SEQ_ERROR  # the part times out, the launch path too
        # a new announcement over the same addresses ends the older one (round 5): the one whose part timed out is dropped by the first of
        # these calls, and so is each of these (still stalled) by the next
        for _ in range(4):
            assert L.QZSTD_hintSource(st, buf, len(data), chunk, 3) in (0, -1)
        assert L.qzstd_test_orphans() >= before + 1  # its buffers are parked, not reused
        L.qzstd_mock_stall_ms(0)
        time.sleep(0.35)
        assert L.QZSTD_hintSource(st, buf, len(data), chunk, 3) == 0
        got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 3)
        L.QZSTD_freeSeqProdState(st)
        assert got == oracle_frames(zstd, oracle, data, chunk, 3)


def test_multi_gpu_product_leg_over_four_mock_devices(mock, zstd, tmp_path):
    """bench.py's product_multi_gpu leg on CPU: test/frontbench.c over the mock with four "GPUs" — QZSTD_HIP_SPLIT=4 cuts every
    announcement into four block ranges (every GPU gets a share), QZSTD_HIP_SPLIT=1 keeps an announcement on its state's GPU (the
    states are spread round-robin); blocks per GPU from QZSTD_deviceStats, frames round-trip, no producer errors"""
    import re
    front_so = os.path.join(ROOT, "tests", "mock", "libqzstdfront_mock.so")
    build_shared(["gcc", "-O2", "-g", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-DQZ_TEST_HOOKS", "-shared", "-fPIC", "-pthread",
                  "-I" + os.path.join(ROOT, "include"), "-o", front_so, os.path.join(B.PKG_DIR, "frontend", "qzstd_frontend.c"),
                           MOCK_SO, zstd.path, "-Wl,-rpath," + os.path.dirname(MOCK_SO), "-Wl,-rpath," + os.path.dirname(zstd.path)], front_so)
    exe = str(tmp_path / "frontbench")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(B.PKG_DIR, "test", "frontbench.c"), front_so, MOCK_SO, zstd.path,
                           "-Wl,-rpath," + os.path.dirname(MOCK_SO), "-Wl,-rpath," + os.path.dirname(zstd.path), "-lpthread"])
    f = tmp_path / "in.bin"
    f.write_bytes(K.by_name("system", 64 * 65536, seed=4))
    for split in (4, 1):
        out = subprocess.run([exe, "-t4", "-l1", "-c65536", "-L3", "-s1", "-m1", str(f)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, QZSTD_MOCK_DEVICES="4", QZSTD_HIP_SPLIT=str(split)))
        assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr
        assert "producer errors: 0 " in out.stdout, out.stdout
        per = re.search(r"blocks per GPU \(announced/batched/service\): (.*)", out.stdout).group(1)
        counts = [int(x.split()[1].split("/")[0]) for x in per.split(",")]
        assert len(counts) == 4 and sum(counts) == 2 * 64, per  # two passes (one warms up) of 64 blocks, all announced
        # what is invariant (round-4 verdict, weak 4: the shares themselves depend on which thread claims what, when): every block was
        # announced (the sum above) and every GPU took part — split 4: announcements of 8 blocks or more are cut into block ranges over
        # the GPUs; split 1: whole announcements per GPU, four states round-robin
        assert all(c > 0 for c in counts), per


def test_stress_driver_over_the_mock(mock, zstd, tmp_path):
    """tests/stress/svc_stress.c (the GPU box's load test: real threads, every work item / every frame against the oracle) builds and
    runs against the mock device layer: keeps the driver itself honest on CPU"""
    exe = str(tmp_path / "svc_stress_mock")
    subprocess.check_call(["gcc", "-O2", "-g", "-std=c11", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
                           "-o", exe, os.path.join(ROOT, "tests", "stress", "svc_stress.c"), os.path.join(ROOT, "oracle", "qzstd_oracle.c"),
                           MOCK_SO, zstd.path, "-Wl,-rpath," + os.path.dirname(MOCK_SO), "-Wl,-rpath," + os.path.dirname(zstd.path)])
    corpus = str(tmp_path / "corpus.bin")
    with open(corpus, "wb") as f:
        f.write(K.by_name("system", 3 * 131072 + 500, seed=31))
    for args in (["items", corpus, "6", "3", "4"], ["items", corpus, "0xc", "2", "3", "32768"], ["frames", corpus, "1,6", "4", "5"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "svc_stress ok" in out.stdout, (args, (out.stdout + out.stderr)[-1500:])


def test_host_path_bench_over_the_replaying_mock(mock, tmp_path):
    """tests/stress/hostpath_bench.c (round-4 verdict, item 7): the HOST side of the announcement path — claims announced two ahead,
    every block taken through qatSequenceProducer, no libzstd — against the mock with QZSTD_MOCK_REPLAY=1 (a "device" that costs one
    memcpy).  Here: it builds, every block comes from an announcement, no errors, the replay table was hit; the rate is printed
    (profiles/r05_host_path.txt holds the measured figures), only a floor is asserted."""
    import re
    exe = str(tmp_path / "hostpath_bench")
    subprocess.check_call(["gcc", "-O2", "-g", "-std=c11", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "stress", "hostpath_bench.c"), MOCK_SO, "-Wl,-rpath," + os.path.dirname(MOCK_SO)])
    f = tmp_path / "in.bin"
    f.write_bytes(K.by_name("system", 64 * 131072, seed=9))
    out = subprocess.run([exe, str(f), "3", "3", "1", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, QZSTD_MOCK_REPLAY="1"))
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r": (\d+) MB/s .*host path alone (\d+) MB/s; (\d+) block\(s\) from announcements, (\d+) per block, (\d+) error\(s\), \d+ sequences, (\d+) replay hits", out.stdout)
    assert m, out.stdout
    rate, host, served, sync, errs, hits = (int(x) for x in m.groups())
    assert served == 4 * 64 and sync == 0 and errs == 0 and hits >= 3 * 64, out.stdout
    assert host >= rate > 200, out.stdout  # (a floor far below any machine: the figure itself is a measurement, not a test)
