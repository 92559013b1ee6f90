"""Repeat-offset aware mode end to end on a GPU box: the caller turns ZSTD_c_searchForExternalRepcodes on
(benchmark tool: -E1) and tells the plugin so with QZSTD_HIP_EXT_REPCODES=1; frames must equal the
ones libzstd makes from the oracle's sequences at level | 0x100, and beat software zstd at level 1."""
import ctypes as C
import os
import subprocess

import pytest

import make_golden as G
import qz_bind as B
import qz_corpus as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def started_rep(gpu_plugin):
    os.environ["QZSTD_HIP_EXT_REPCODES"] = "1"
    try:
        assert gpu_plugin.lib.QZSTD_startQatDevice() == 0
        yield gpu_plugin
        gpu_plugin.lib.QZSTD_stopQatDevice()
    finally:
        del os.environ["QZSTD_HIP_EXT_REPCODES"]


def plugin_frames(zstd, plug, data, chunk, level, hint=False):
    st = plug.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(level, producer=plug.producer_addr, state=st, fallback=False, validate=True, ext_repcodes=1)
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    if hint:
        assert plug.lib.QZSTD_hintSource(st, buf, len(data), chunk, level) == 0
    cap = zstd.lib.ZSTD_compressBound(chunk)
    dst = C.create_string_buffer(cap)
    frames = []
    for o in range(0, len(data), chunk):
        r = zstd.lib.ZSTD_compress2(zc, dst, cap, C.byref(buf, o), min(chunk, len(data) - o))
        assert not zstd.is_error(r), zstd.err(r)
        frames.append(dst.raw[:r])
    zstd.free(zc)
    plug.lib.QZSTD_freeSeqProdState(st)
    return frames


@pytest.mark.parametrize("level,chunk,hint", [(1, 131072, False), (1, 131072, True), (3, 65536, False), (6, 131072, True),
                                              (12, 32768, False)])
def test_frames_equal_oracle_frames(started_rep, zstd, oracle, level, chunk, hint):
    data = K.by_name("system", 10 * 131072 + 4321)
    got = plugin_frames(zstd, started_rep, data, chunk, level, hint)
    zc, keep = G.oracle_cctx(zstd, oracle, level | 0x100, chunk)
    _, want = zstd.compress_chunks(zc, data, chunk)
    zstd.free(zc)
    assert got == want
    assert b"".join(zstd.decompress(f, chunk) for f in got) == data


def test_level1_beats_software_with_repcodes(started_rep, zstd):
    data = K.by_name("system", 48 * 131072)
    ours = sum(len(f) for f in plugin_frames(zstd, started_rep, data, 131072, 1, hint=True))
    zc = zstd.cctx(1)
    sw, _ = zstd.compress_chunks(zc, data, 131072)
    zstd.free(zc)
    assert ours <= sw, (ours, sw)


def test_benchmark_tool_with_E1(started_rep, tmp_path):
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "benchmark", "ZSTDLIB=" + B.find_libzstd()], stdout=subprocess.DEVNULL)
    f = tmp_path / "corpus.bin"
    f.write_bytes(K.by_name("system", 24 * 131072))
    env = dict(os.environ, QZSTD_HIP_EXT_REPCODES="1")
    out = subprocess.run([os.path.join(tdir, "benchmark"), "-m1", "-t2", "-l1", "-c128K", "-L3", "-E1", "-H1", str(f)],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    assert out.stderr.count("PASS") == 2, out.stderr
