"""The drop-in surface on a GPU box: ZSTD_compress2 / ZSTD_compressStream2 with
qatSequenceProducer registered must (a) round-trip bit-exactly, (b) produce exactly the
frame libzstd produces from the oracle's sequences, (c) stay within 2 % of libzstd's own
software match-finder at the same level (benchmark.c framing: one frame per chunk)."""
import ctypes as C
import threading

import pytest

import qz_bind as B
import qz_corpus as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def started(gpu_plugin):
    assert gpu_plugin.lib.QZSTD_startQatDevice() == 0  # QZSTD_OK
    yield gpu_plugin
    gpu_plugin.lib.QZSTD_stopQatDevice()


def compress_with(zstd, producer_addr, state, data, chunk, level, hint_lib=None):
    zc = zstd.cctx(level, producer=producer_addr, state=state, fallback=False, validate=True)
    if hint_lib is not None:
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        assert hint_lib.QZSTD_hintSource(state, buf, len(data), chunk, level) == 0
        frames = []
        cap = zstd.lib.ZSTD_compressBound(chunk)
        dst = C.create_string_buffer(cap)
        for o in range(0, len(data), chunk):
            n = min(chunk, len(data) - o)
            r = zstd.lib.ZSTD_compress2(zc, dst, cap, C.byref(buf, o), n)
            assert not zstd.is_error(r), zstd.err(r)
            frames.append(dst.raw[:r])
    else:
        _, frames = zstd.compress_chunks(zc, data, chunk)
    zstd.free(zc)
    return frames


@pytest.mark.parametrize("level,chunk", [(1, 131072), (3, 131072), (6, 131072), (12, 32768), (1, 65536)])
def test_compress2_matches_oracle_frames(started, zstd, oracle, level, chunk):
    data = K.mix(21, 12 * 131072 + 4321)
    st = started.lib.QZSTD_createSeqProdState()
    got = compress_with(zstd, started.producer_addr, st, data, chunk, level)
    started.lib.QZSTD_freeSeqProdState(st)
    want = compress_with(zstd, oracle.producer_addr, None, data, chunk, level)
    assert got == want  # same sequences in -> same frame bytes out
    out = b"".join(zstd.decompress(f, chunk) for f in got)
    assert out == data


def test_hint_batch_path_identical(started, zstd, oracle):
    data = K.by_name("system", 24 * 131072 + 777)
    st = started.lib.QZSTD_createSeqProdState()
    got = compress_with(zstd, started.producer_addr, st, data, 131072, 1, hint_lib=started.lib)
    started.lib.QZSTD_freeSeqProdState(st)
    want = compress_with(zstd, oracle.producer_addr, None, data, 131072, 1)
    assert got == want
    assert b"".join(zstd.decompress(f, 131072) for f in got) == data


@pytest.mark.parametrize("level,chunk,seg", [(1, 131072, 8), (3, 65536, 5), (1, 32768, 16)])
def test_hint_double_buffered_lookahead(started, zstd, oracle, level, chunk, seg):
    """segment k+1 is announced (asynchronous launch) before segment k is compressed: frames must
    still be exactly the oracle's, including the ragged last segment"""
    data = K.by_name("system", 5 * seg * chunk + 3 * chunk + 999)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    segb = seg * chunk
    st = started.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(level, producer=started.producer_addr, state=st, fallback=False, validate=True)
    cap = zstd.lib.ZSTD_compressBound(chunk)
    dst = C.create_string_buffer(cap)
    assert started.lib.QZSTD_hintSource(st, buf, min(segb, len(data)), chunk, level) == 0
    got = []
    for o in range(0, len(data), chunk):
        if o % segb == 0 and o + segb < len(data):
            n2 = min(segb, len(data) - o - segb)
            assert started.lib.QZSTD_hintSource(st, C.byref(buf, o + segb), n2, chunk, level) == 0
        n = min(chunk, len(data) - o)
        r = zstd.lib.ZSTD_compress2(zc, dst, cap, C.byref(buf, o), n)
        assert not zstd.is_error(r), zstd.err(r)
        got.append(dst.raw[:r])
    zstd.free(zc)
    started.lib.QZSTD_freeSeqProdState(st)
    want = compress_with(zstd, oracle.producer_addr, None, data, chunk, level)
    assert got == want
    assert b"".join(zstd.decompress(f, chunk) for f in got) == data


def test_hint_limits_and_abandoned_hints(started, zstd):
    """oversized announcements are refused; announcements that are never consumed must give their
    slot back when the state is freed (more states than slots, then a normal compression)"""
    L = started.lib
    st = L.QZSTD_createSeqProdState()
    big = C.create_string_buffer((16 << 20) + 131072)
    assert L.QZSTD_hintSource(st, big, len(big), 131072, 1) == -1
    assert L.QZSTD_hintSource(st, big, 1 << 20, 131072 + 16, 1) == -1   # block grid > 128 KiB
    assert L.QZSTD_hintSource(st, big, 1 << 20, 1000, 1) == -1           # grid not 16-aligned
    assert L.QZSTD_hintSource(st, big, 1 << 20, 131072, 0) == -1         # level guard
    L.QZSTD_freeSeqProdState(st)
    data = K.text(3, 4 * 131072)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    for _ in range(80):
        s2 = L.QZSTD_createSeqProdState()
        assert L.QZSTD_hintSource(s2, buf, len(data), 131072, 1) == 0
        assert L.QZSTD_hintSource(s2, buf, len(data), 65536, 1) == 0
        assert L.QZSTD_hintSource(s2, buf, len(data), 131072, 2) == 0    # re-uses (and drains) buffer 0
        L.QZSTD_freeSeqProdState(s2)
    st = L.QZSTD_createSeqProdState()
    frames = compress_with(zstd, started.producer_addr, st, data, 131072, 1)
    L.QZSTD_freeSeqProdState(st)
    assert b"".join(zstd.decompress(f, 131072) for f in frames) == data


def test_one_shot_multiblock_frame(started, zstd):
    """plain ZSTD_compress2 over 1 MiB+: libzstd calls the producer once per 128 KiB block
    (and 1.5.7 may pre-split); every call is an independent block."""
    data = K.text(8, (1 << 20) + 300000)
    st = started.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(1, producer=started.producer_addr, state=st, fallback=False, validate=True)
    frame = zstd.compress2(zc, data)
    zstd.free(zc)
    started.lib.QZSTD_freeSeqProdState(st)
    assert zstd.decompress(frame, len(data)) == data


@pytest.mark.parametrize("level,chunk,corpus,size", [(1, 131072, "system", 64 * 131072), (3, 131072, "system", 64 * 131072),
                                                     (6, 131072, "system", 48 * 131072), (6, 131072, "text", 32 * 131072),
                                                     (8, 131072, "system", 24 * 131072), (9, 131072, "system", 24 * 131072),
                                                     (11, 131072, "system", 24 * 131072),
                                                     (12, 32768, "weblog", 96 * 32768), (12, 32768, "system", 96 * 32768)])
def test_ratio_within_2pct_of_software(started, zstd, level, chunk, corpus, size):
    """north star: compressed size within 2 % of libzstd's own match-finder at the same level, same framing (one frame
    per chunk, reference test/benchmark.c:300-321), through the real callback path with libzstd's default parameters.
    Level 6 on 128 KiB text blocks = BASELINE config 3, level 12 on 32 KiB web-log blocks = config 4."""
    data = K.by_name(corpus, size, seed={"text": 3, "weblog": 4}.get(corpus, 1))
    st = started.lib.QZSTD_createSeqProdState()
    got = compress_with(zstd, started.producer_addr, st, data, chunk, level, hint_lib=started.lib)
    started.lib.QZSTD_freeSeqProdState(st)
    assert b"".join(zstd.decompress(f, chunk) for f in got) == data
    zc = zstd.cctx(level)
    sw, _ = zstd.compress_chunks(zc, data, chunk)
    zstd.free(zc)
    ours = sum(len(f) for f in got)
    assert ours <= sw * 1.02, "level %d: compressed %d vs software %d (%.2f %% worse)" % (level, ours, sw, 100.0 * (ours / sw - 1))


def test_guards_match_reference(started):
    """argument guards of reference src/qatseqprod.c:1123-1137"""
    L = started.lib
    st = L.QZSTD_createSeqProdState()
    seqs = (B.Sequence * 64)()
    src = C.create_string_buffer(K.incompressible(1, 100))
    err = B.SEQ_ERROR
    assert L.qatSequenceProducer(st, seqs, 64, src, 100, None, 0, 0, 1 << 17) == err      # level 0
    assert L.qatSequenceProducer(st, seqs, 64, src, 100, None, 0, 13, 1 << 17) == err     # level 13
    assert L.qatSequenceProducer(st, seqs, 64, src, 100, src, 0, 1, 1 << 17) == err       # dict pointer
    assert L.qatSequenceProducer(st, seqs, 64, src, 100, None, 5, 1, 1 << 17) == err      # dict size
    assert L.qatSequenceProducer(st, seqs, 64, src, 100, None, 0, 1, 50) == err           # window < srcSize
    assert L.qatSequenceProducer(st, seqs, 64, src, 100, None, 0, 1, 100) == 1            # incompressible -> 1 delimiter
    assert (seqs[0].offset, seqs[0].litLength, seqs[0].matchLength) == (0, 100, 0)
    L.QZSTD_freeSeqProdState(st)


def test_threads_each_with_own_cctx(started, zstd, oracle):
    """reference scaling model: one CCtx + one state per thread (README.md:138, benchmark.c:514-516)"""
    data = [K.mix(100 + t, 6 * 131072) for t in range(6)]
    res = [None] * len(data)

    def work(t):
        z = B.Zstd(zstd.path)
        st = started.lib.QZSTD_createSeqProdState()
        res[t] = compress_with(z, started.producer_addr, st, data[t], 131072, 1)
        started.lib.QZSTD_freeSeqProdState(st)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(len(data))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for t in range(len(data)):
        assert res[t] == compress_with(zstd, oracle.producer_addr, None, data[t], 131072, 1)


def test_compress_stream2_chunked_frames(started, zstd):
    """BASELINE config #5 shape at test scale: ZSTD_compressStream2, 4 MiB frames (ZSTD_e_end per
    frame), level 3: src then points into libzstd's own buffer, blocks are still independent."""
    import ctypes as C
    L = zstd.lib

    class InB(C.Structure):
        _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    class OutB(C.Structure):
        _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(OutB), C.POINTER(InB), C.c_int]
    L.ZSTD_compressStream2.restype = C.c_size_t
    data = K.mixed_entropy(5, 2 * (4 << 20) + 70000)
    st = started.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(3, producer=started.producer_addr, state=st, fallback=False, validate=True)
    frames = []
    for o in range(0, len(data), 4 << 20):
        part = data[o:o + (4 << 20)]
        src = C.create_string_buffer(part, len(part))
        cap = L.ZSTD_compressBound(len(part))
        dst = C.create_string_buffer(cap)
        out = OutB(C.cast(dst, C.c_void_p), cap, 0)
        pos = 0
        while True:  # feed in 256 KiB pieces
            n = min(256 << 10, len(part) - pos)
            inp = InB(C.cast(C.byref(src, pos), C.c_void_p), n, 0)
            last = pos + n >= len(part)
            r = L.ZSTD_compressStream2(zc, C.byref(out), C.byref(inp), B.e_end if last else B.e_continue)
            assert not zstd.is_error(r), zstd.err(r)
            pos += inp.pos
            if last and r == 0:
                break
        frames.append((dst.raw[:out.pos], len(part)))
    zstd.free(zc)
    started.lib.QZSTD_freeSeqProdState(st)
    assert b"".join(zstd.decompress(f, n) for f, n in frames) == data


def test_config5_shape_stream_frames_with_lookahead(started, zstd, oracle):
    """BASELINE config #5 in small: mixed-entropy data fed to ZSTD_compressStream2 as 4 MiB frames
    (one call with ZSTD_e_end per frame, level 3), the next frame announced while the current one is
    compressed.  With the whole frame handed over at once libzstd reads the caller's buffer directly, so
    every one of the 32 callbacks per frame is served from the announcement (QZSTD_hintStats), and the
    frames equal the oracle's.  libzstd 1.5.7 pre-splits the blocks of multi-block frames from level 3 on
    (callbacks of ~30 KiB that no longer sit on a block grid and take the per-block path);
    ZSTD_c_blockSplitterLevel = 1 switches that off."""
    L = zstd.lib

    class InB(C.Structure):
        _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    class OutB(C.Structure):
        _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(OutB), C.POINTER(InB), C.c_int]
    L.ZSTD_compressStream2.restype = C.c_size_t
    frame, nframes, level = 4 << 20, 5, 3
    data = K.mixed_entropy(5, nframes * frame)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    cap = L.ZSTD_compressBound(frame)
    dst = C.create_string_buffer(cap)

    def run(producer, state, hints, split_off=True):
        zc = zstd.cctx(level, producer=producer, state=state, fallback=False, validate=True,
                       **({"blockSplitterLevel": 1} if split_off else {}))
        frames = []
        if hints:
            assert started.lib.QZSTD_hintSource(state, buf, frame, 131072, level) == 0
        for f in range(nframes):
            if hints and f + 1 < nframes:
                assert started.lib.QZSTD_hintSource(state, C.byref(buf, (f + 1) * frame), frame, 131072, level) == 0
            out = OutB(C.cast(dst, C.c_void_p), cap, 0)
            inp = InB(C.cast(C.byref(buf, f * frame), C.c_void_p), frame, 0)
            r = L.ZSTD_compressStream2(zc, C.byref(out), C.byref(inp), B.e_end)
            assert r == 0 and inp.pos == frame, zstd.err(r)
            frames.append(dst.raw[:out.pos])
        zstd.free(zc)
        return frames

    st = started.lib.QZSTD_createSeqProdState()
    got = run(started.producer_addr, st, True)
    stats = (C.c_ulong * 4)()
    started.lib.QZSTD_hintStats(st, C.byref(stats))
    started.lib.QZSTD_freeSeqProdState(st)
    want = run(oracle.producer_addr, None, False)
    assert got == want
    assert b"".join(zstd.decompress(f, frame) for f in got) == data
    assert stats[2] == nframes and stats[1] == 0 and stats[0] == nframes * 32, list(stats)
    # with libzstd's pre-splitter left on, the output is still right (per-block path)
    st = started.lib.QZSTD_createSeqProdState()
    got = run(started.producer_addr, st, True, split_off=False)
    started.lib.QZSTD_freeSeqProdState(st)
    assert b"".join(zstd.decompress(f, frame) for f in got) == data


def test_benchmark_tool_gpu_modes(started, tmp_path):
    """the C benchmark (counterpart of reference test/benchmark.c) with the producer registered"""
    import os
    import subprocess
    zpath = B.find_libzstd()
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "benchmark", "ZSTDLIB=" + zpath], stdout=subprocess.DEVNULL)
    f = tmp_path / "corpus.bin"
    f.write_bytes(K.by_name("system", 24 * 131072))
    exe = os.path.join(tdir, "benchmark")
    for extra in ([], ["-H1"]):
        out = subprocess.run([exe, "-m1", "-t3", "-l2", "-c128K", "-L1"] + extra + [str(f)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stderr.count("PASS") == 3, out.stderr


@pytest.mark.parametrize("level", [1, 6])
def test_per_slot_path_without_coalescing(started, tmp_path, level):
    """QZSTD_HIP_COALESCE=0: every caller owns a slot with its own stream and buffers for the duration of a
    call, the reference's instance model (src/qatseqprod.c:905-933, :1156-1334); level 6 also exercises the
    per-slot chain workspace"""
    import os
    import subprocess
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "benchmark", "ZSTDLIB=" + B.find_libzstd()], stdout=subprocess.DEVNULL)
    f = tmp_path / "corpus.bin"
    f.write_bytes(K.by_name("system", 16 * 131072 + 4567))
    env = dict(os.environ, QZSTD_HIP_COALESCE="0", QZSTD_HIP_SLOTS="4")
    out = subprocess.run([os.path.join(tdir, "benchmark"), "-m1", "-t6", "-l1", "-c128K", "-L%d" % level, str(f)],
                         capture_output=True, text=True, env=env)  # more threads than slots: callers wait for one
    assert out.returncode == 0, out.stderr
    assert out.stderr.count("PASS") == 6, out.stderr


def test_coalescer_stress_mixed_levels_and_sizes(started, zstd, oracle):
    """many callers, two levels, ragged chunk sizes: the group-commit coalescer must hand every
    caller exactly its own block's sequences (frames identical to the oracle's)"""
    import random
    nthreads = 24
    jobs = []
    rng = random.Random(7)
    for t in range(nthreads):
        level = 1 if t % 3 else 3
        chunk = rng.choice([131072, 65536, 100000, 32768, 4096])
        data = K.by_name(rng.choice(["text", "binary", "weblog", "mix"]), chunk * 5 + rng.randrange(1, 3000), seed=200 + t)
        jobs.append((level, chunk, data))
    res = [None] * nthreads

    def work(t):
        level, chunk, data = jobs[t]
        z = B.Zstd(zstd.path)
        st = started.lib.QZSTD_createSeqProdState()
        res[t] = compress_with(z, started.producer_addr, st, data, chunk, level)
        started.lib.QZSTD_freeSeqProdState(st)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for t in range(nthreads):
        level, chunk, data = jobs[t]
        assert res[t] is not None
        assert res[t] == compress_with(zstd, oracle.producer_addr, None, data, chunk, level), "thread %d" % t


def test_hinting_and_plain_callers_side_by_side(started, zstd, oracle):
    """threads that announce their buffers (slots held while announcements are in flight) next to threads that
    do not (coalescer), at a chain level and a plain one: everybody gets exactly the oracle's frames"""
    nthreads = 12
    jobs = []
    for t in range(nthreads):
        level = 6 if t % 4 == 0 else 1
        chunk = 131072 if t % 3 else 65536  # both are 16-aligned block grids
        data = K.by_name(["text", "binary", "weblog", "mix"][t % 4], chunk * 9 + 17 * t, seed=300 + t)
        jobs.append((level, chunk, data, t % 2 == 0))
    res = [None] * nthreads

    def work(t):
        level, chunk, data, hint = jobs[t]
        z = B.Zstd(zstd.path)
        st = started.lib.QZSTD_createSeqProdState()
        res[t] = compress_with(z, started.producer_addr, st, data, chunk, level, hint_lib=started.lib if hint else None)
        started.lib.QZSTD_freeSeqProdState(st)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for t in range(nthreads):
        level, chunk, data, _ = jobs[t]
        assert res[t] is not None, "thread %d died" % t
        assert res[t] == compress_with(zstd, oracle.producer_addr, None, data, chunk, level), "thread %d" % t


def _compress_chunks_raw(zstd, plug, st, buf_addr, total, chunk, level, order=None, before=None):
    """one ZSTD_compress2 per chunk straight from memory at buf_addr (no copies: the producer sees the real addresses)"""
    zc = zstd.cctx(level, producer=plug.producer_addr, state=st, fallback=False, validate=True)
    cap = zstd.lib.ZSTD_compressBound(chunk)
    dst = C.create_string_buffer(cap)
    frames = {}
    idx = list(range((total + chunk - 1) // chunk))
    for c in (order or idx):
        if before:
            before(c)
        n = min(chunk, total - c * chunk)
        r = zstd.lib.ZSTD_compress2(zc, dst, cap, C.c_void_p(buf_addr + c * chunk), n)
        assert not zstd.is_error(r), zstd.err(r)
        frames[c] = dst.raw[:r]
    zstd.free(zc)
    return [frames[c] for c in idx]


def test_unchanged_callers_read_nothing_but_the_callbacks_block(started, zstd, oracle):
    """no hints: every block takes the per-block path (the resident service), and the library reads [src, src + srcSize) of a callback
    and nothing else — the buffer ends right in front of an unreadable page, the caller rewrites every chunk just before compressing
    it and jumps around; QZSTD_HIP_LOOKAHEAD (the opt-in guessing of rounds 1-4, removed: round-4 verdict item 6) is ignored"""
    import mmap
    import os
    import random
    libc = C.CDLL(None, use_errno=True)
    page = mmap.PAGESIZE
    nblk, chunk = 12, 65536
    total = nblk * chunk
    mm = mmap.mmap(-1, total + page)
    base = C.addressof(C.c_char.from_buffer(mm))
    assert libc.mprotect(C.c_void_p(base + total), C.c_size_t(page), 0) == 0  # PROT_NONE right behind the data
    final = K.by_name("mix", total, seed=77)
    mm[:total] = K.by_name("text", total, seed=5)

    def rewrite(c):
        mm[c * chunk:(c + 1) * chunk] = final[c * chunk:(c + 1) * chunk]

    stats = (C.c_ulong * 4)()
    started.lib.QZSTD_stopQatDevice()
    os.environ["QZSTD_HIP_LOOKAHEAD"] = "1"
    try:
        assert started.lib.QZSTD_startQatDevice() == 0
        st = started.lib.QZSTD_createSeqProdState()
        got = _compress_chunks_raw(zstd, started, st, base, total, chunk, 1, before=rewrite)
        assert got == compress_with(zstd, oracle.producer_addr, None, final, chunk, 1)
        order = list(range(nblk))
        random.Random(3).shuffle(order)
        got = _compress_chunks_raw(zstd, started, st, base, total, chunk, 3, order=order)
        started.lib.QZSTD_hintStats(st, C.byref(stats))
        started.lib.QZSTD_freeSeqProdState(st)
    finally:
        del os.environ["QZSTD_HIP_LOOKAHEAD"]
        started.lib.QZSTD_stopQatDevice()
        assert started.lib.QZSTD_startQatDevice() == 0
    assert got == compress_with(zstd, oracle.producer_addr, None, final, chunk, 3)
    assert stats[0] == 0 and stats[1] == 2 * nblk and stats[2] == 0, list(stats)  # all per block, nothing announced or guessed
    assert libc.mprotect(C.c_void_p(base + total), C.c_size_t(page), 3) == 0
    del got


def test_callbacks_spanning_several_grid_blocks(started, zstd, oracle):
    """an announcement on a 64 KiB grid serves 128 KiB callbacks too (what libzstd 1.5.7 mixes inside multi-block
    frames): the two independently parsed halves are concatenated, the first half's trailing literals flowing
    into the second half's first sequence.  Every callback is served from the announcement; the frames decode
    exactly and are what libzstd makes of the oracle's two half-block parses joined the same way."""
    data = K.by_name("system", 10 * 131072)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    st = started.lib.QZSTD_createSeqProdState()
    assert started.lib.QZSTD_hintSource(st, buf, len(data), 65536, 1) == 0
    got = _compress_chunks_raw(zstd, started, st, C.addressof(buf), len(data), 131072, 1)
    stats = (C.c_ulong * 4)()
    started.lib.QZSTD_hintStats(st, C.byref(stats))
    started.lib.QZSTD_freeSeqProdState(st)
    assert b"".join(zstd.decompress(f, 131072) for f in got) == data
    assert stats[0] == 10 and stats[1] == 0, list(stats)

    # the same join done by hand on the oracle's sequences, fed to libzstd through a tiny replay producer
    prof = oracle.profile(1, 65536)
    joined = {}
    for c in range(10):
        blk = data[c * 131072:(c + 1) * 131072]
        n1, s1 = oracle.find(prof, blk[:65536])
        n2, s2 = oracle.find(prof, blk[65536:])
        a = [(s1[i].offset, s1[i].litLength, s1[i].matchLength) for i in range(n1)]
        b2 = [(s2[i].offset, s2[i].litLength, s2[i].matchLength) for i in range(n2)]
        carry = a[-1][1]
        seqs = a[:-1]
        if len(b2) > 1:
            seqs += [(b2[0][0], b2[0][1] + carry, b2[0][2])] + b2[1:-1]
            carry = 0
        seqs.append((0, carry + b2[-1][1], 0))
        joined[C.addressof(buf) + c * 131072] = seqs

    def replay(state, out, cap, src, n, d, ds, level, win):
        seqs = joined[src]
        for i, (o, l, m) in enumerate(seqs):
            out[i].offset, out[i].litLength, out[i].matchLength, out[i].rep = o, l, m, 0
        return len(seqs)

    cb = B.PRODUCER_F(replay)
    want = _compress_chunks_raw(zstd, type("P", (), {"producer_addr": C.cast(cb, C.c_void_p)})(), None, C.addressof(buf), len(data),
                                131072, 1)
    assert got == want


def test_streaming_caller_is_served_from_an_announcement_by_content(started, zstd):
    """ZSTD_compressStream2 fed 50 000 bytes at a time from a different buffer than the announced one: libzstd's blocks come
    out of its own window buffer, the announcement is matched by content (fingerprint + memcmp)"""
    L = zstd.lib

    class InB(C.Structure):
        _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    class OutB(C.Structure):
        _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(OutB), C.POINTER(InB), C.c_int]
    L.ZSTD_compressStream2.restype = C.c_size_t
    data = K.by_name("system", 24 * 131072 + 777)
    announced = (C.c_char * len(data)).from_buffer_copy(data)
    feed = (C.c_char * len(data)).from_buffer_copy(data)
    st = started.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(3, producer=started.producer_addr, state=st, fallback=False, validate=True, blockSplitterLevel=1)
    assert started.lib.QZSTD_hintSource(st, announced, len(data), 131072, 3) == 0
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
    stats = (C.c_ulong * 4)()
    started.lib.QZSTD_hintStats(st, C.byref(stats))
    zstd.free(zc)
    started.lib.QZSTD_freeSeqProdState(st)
    assert zstd.decompress(dst.raw[:out.pos], len(data)) == data
    assert stats[0] >= 24 and stats[0] + stats[1] == 25, list(stats)
