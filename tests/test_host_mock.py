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


@pytest.fixture(scope="module")
def mock(oracle):
    srcs = [os.path.join(B.PKG_DIR, "host", "qatseqprod.c"), os.path.join(B.PKG_DIR, "csrc", "qzstd_profile.c"),
            os.path.join(ROOT, "tests", "mock", "mock_hip.c"), os.path.join(ROOT, "oracle", "qzstd_oracle.c")]
    subprocess.check_call(["gcc", "-O2", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-shared", "-fPIC", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-o", MOCK_SO] + srcs)
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
def test_unchanged_caller_guessed_lookahead(mock, zstd, oracle, level, chunk):
    data = K.by_name("mix", 24 * chunk + 777, seed=level)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    st = mock.lib.QZSTD_createSeqProdState()
    got = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, level)
    stats = stats_of(mock, st)
    mock.lib.QZSTD_freeSeqProdState(st)
    assert got == oracle_frames(zstd, oracle, data, chunk, level)
    assert stats[2] == 0 and stats[0] >= 16, stats  # nothing announced; most blocks came from guesses


@pytest.mark.parametrize("mode", ["1", "2"])  # fault-safe read by process_vm_readv / through a pipe
def test_stale_guess_and_unreadable_page(mock, zstd, oracle, mode):
    with restarted(mock, QZSTD_HIP_LOOKAHEAD=mode):
        _stale_guess_and_unreadable_page(mock, zstd, oracle)


def _stale_guess_and_unreadable_page(mock, zstd, oracle):
    libc = C.CDLL(None, use_errno=True)
    page, nblk, chunk = mmap.PAGESIZE, 10, 65536
    total = nblk * chunk
    mm = mmap.mmap(-1, total + page)
    base = C.addressof(C.c_char.from_buffer(mm))
    assert libc.mprotect(C.c_void_p(base + total), C.c_size_t(page), 0) == 0
    final = K.by_name("text", total, seed=11)
    mm[:total] = K.by_name("binary", total, seed=12)

    def rewrite(c):  # the caller produces every chunk only just before compressing it
        mm[c * chunk:(c + 1) * chunk] = final[c * chunk:(c + 1) * chunk]

    st = mock.lib.QZSTD_createSeqProdState()
    got = frames_of(zstd, mock.producer_addr, st, base, total, chunk, 1, before=rewrite)
    assert got == oracle_frames(zstd, oracle, final, chunk, 1)
    got = frames_of(zstd, mock.producer_addr, st, base, total, chunk, 1, order=[7, 2, 9, 0, 1, 3, 8, 4, 6, 5])
    assert got == oracle_frames(zstd, oracle, final, chunk, 1)
    mock.lib.QZSTD_freeSeqProdState(st)
    st = mock.lib.QZSTD_createSeqProdState()  # a fresh state (the old one has backed off): in order, stable content
    got = frames_of(zstd, mock.producer_addr, st, base, total, chunk, 1)
    served = stats_of(mock, st)[0]
    mock.lib.QZSTD_freeSeqProdState(st)
    assert served >= 5, served  # guesses are used right up to the unreadable page
    assert got == oracle_frames(zstd, oracle, final, chunk, 1)
    assert libc.mprotect(C.c_void_p(base + total), C.c_size_t(page), 3) == 0
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
    """QZSTD_HIP_COALESCE=0 (a slot per caller, fewer slots than threads), QZSTD_HIP_LOOKAHEAD=0,
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
    for env in ({"QZSTD_HIP_COALESCE": "0", "QZSTD_HIP_SLOTS": "3"}, {"QZSTD_HIP_LOOKAHEAD": "0"}, {"QZSTD_HIP_LOOKAHEAD": "2"},
                {"QZSTD_HIP_EXT_REPCODES": "1"}):
        out = subprocess.run(["python", "-c", script], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
        assert out.returncode == 0 and "OK" in out.stdout, (env, out.stderr[-800:])


def test_lookahead_never_changes_the_output(mock, zstd):
    """multi-block frames (libzstd 1.5.7 cuts them into irregular 32-128 KiB blocks): byte-identical frames with the
    transparent look-ahead on and off"""
    data = K.by_name("system", 3 * (1 << 20) + 4321)
    buf = (C.c_char * len(data)).from_buffer_copy(data)

    def run(chunk):
        st = mock.lib.QZSTD_createSeqProdState()
        fr = frames_of(zstd, mock.producer_addr, st, C.addressof(buf), len(data), chunk, 1)
        mock.lib.QZSTD_freeSeqProdState(st)
        return fr

    on = {c: run(c) for c in (1 << 20, 393216)}
    with restarted(mock, QZSTD_HIP_LOOKAHEAD="0"):
        off = {c: run(c) for c in (1 << 20, 393216)}
    with restarted(mock, QZSTD_HIP_LOOKAHEAD="2"):
        piped = {c: run(c) for c in (1 << 20, 393216)}
    assert on == off == piped
    assert b"".join(zstd.decompress(f, 1 << 20) for f in on[1 << 20]) == data
