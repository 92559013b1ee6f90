"""CPU: tools/zstdshim — the normally-built zstd 1.5.7 inside pyarrow's libarrow.so behind the public ZSTD_* names (zstdshim.c says why).
It is the libzstd of every test, tool and bench leg when it works (tools/qz_bind.py: find_libzstd); these tests pin what "works" means:
every public name resolved, the same version as the image's exported copy (Pillow's, a ~4x slower build), BYTE-IDENTICAL frames from both
— software levels and through the sequence-producer API with the oracle as producer — and a faster entropy stage."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

import qz_bind as B
import qz_corpus as K


@pytest.fixture(scope="module")
def shim():
    path = B.fast_libzstd()
    if not path:
        pytest.skip("tools/zstdshim unusable here (no gcc, no pyarrow, or QZ_ZSTD_NO_SHIM=1): the suite runs on the exported libzstd")
    return path


def test_every_public_name_resolved_from_libarrow(shim):
    L = C.CDLL(shim)
    L.zstdshim_source.restype = C.c_char_p
    assert L.zstdshim_ok() == 1
    assert L.zstdshim_resolved() == L.zstdshim_wanted() >= 170
    assert b"libarrow.so" in L.zstdshim_source()
    names = [ln[2:-1] for ln in open(os.path.join(B.SHIM_DIR, "zstd_names.h")).read().split("\n") if ln.startswith("X(")]
    assert len(names) == L.zstdshim_wanted() and "ZSTD_registerSequenceProducer" in names and "ZSTD_compress2" in names
    for n in names:
        assert hasattr(L, n), n
    L.ZSTD_versionNumber.restype = C.c_uint
    assert L.ZSTD_versionNumber() >= 10504
    assert B.find_libzstd() == shim or os.environ.get("ZSTDLIB")


CHILD = r"""
import hashlib, json, sys, time
sys.path.insert(0, %r)
import qz_bind as B, qz_corpus as K
path, what = sys.argv[1], sys.argv[2]
z = B.Zstd(path)   # ONE libzstd per process: two in one process interpose each other's symbols (both are loaded RTLD_GLOBAL)
out = {"version": z.version()}
if what == "frames":
    orc = B.Oracle()
    data = K.by_name("system", 6 * 131072 + 999, seed=21) + K.by_name("weblog", 2 * 131072, seed=2)
    for level in (1, 3, 6):
        c = z.cctx(level)
        out["sw%%d" %% level] = hashlib.sha256(b"".join(z.compress_chunks(c, data, 131072)[1])).hexdigest()
        z.free(c)
    for level, chunk in ((1, 131072), (6, 131072), (12, 32768)):
        c = z.cctx(level, producer=orc.producer_addr, state=None, fallback=False, validate=True)
        frames = z.compress_chunks(c, data, chunk)[1]
        z.free(c)
        assert b"".join(z.decompress(f, chunk) for f in frames) == data
        out["producer%%d" %% level] = hashlib.sha256(b"".join(frames)).hexdigest()
else:
    data = K.by_name("system", 64 * 131072, seed=5)
    c = z.cctx(1)
    z.compress_chunks(c, data[:8 * 131072], 131072)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); z.compress_chunks(c, data, 131072); best = min(best, time.perf_counter() - t0)
    out["MBps"] = len(data) / best / 1e6
print(json.dumps(out))
""" % os.path.join(B.ROOT, "tools")


def child(path, what):
    r = subprocess.run([sys.executable, "-c", CHILD, path, what], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_same_frames_as_the_exported_copy(shim, oracle):
    """same version, same decisions: software frames at levels 1/3/6, and frames through the producer API (the oracle as the producer) —
    each library in its own process (two libzstd in one process interpose each other's symbols)"""
    slow_path = B.slow_libzstd()
    if os.path.abspath(slow_path) == os.path.abspath(shim):
        pytest.skip("no second libzstd >= 1.5.4 to compare with")
    a, b = child(shim, "frames"), child(slow_path, "frames")
    if a["version"] != b["version"]:
        pytest.skip("the exported copy is %s, the one inside libarrow.so %s" % (b["version"], a["version"]))
    assert a == b


def test_the_point_of_it_a_faster_library(shim):
    """level 1 on 128 KiB chunks, one thread: the copy inside libarrow.so is several times faster than Pillow's (measured 684 vs 169 MB/s);
    asserted loosely (>= 1.5x) — other tests share the cores"""
    slow_path = B.slow_libzstd()
    if "pillow.libs" not in slow_path:
        pytest.skip("the exported libzstd here is not Pillow's slow build")
    fast, slow = child(shim, "rate")["MBps"], child(slow_path, "rate")["MBps"]
    assert fast >= 1.5 * slow, (fast, slow)


def test_front_end_binds_to_its_own_libzstd_when_another_one_is_preloaded():
    """under a tool that preloads libraries (rocprofv3 brings the system's libzstd 1.4.8 in through libdw) another libzstd sits in front of
    the one Zstd() loads; the front-end library's ZSTD_* references would bind to it and QZSTD_createFront fail on a parameter 1.4.8 does
    not know.  tools/qz_bind.py: Front() notices and binds the library to its own dependencies first (RTLD_DEEPBIND).  On CPU the producer
    reports "device down" and libzstd falls back to its own match-finder: the frames still have to round-trip"""
    old = [p for p in ("/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/lib/x86_64-linux-gnu/libzstd.so.1") if os.path.exists(p)]
    if not old or not os.path.isfile(B.FRONT_SO):
        pytest.skip("no second libzstd to preload, or the front-end library is not built")
    code = r"""
import sys; sys.path.insert(0, %r)
import qz_bind as B, qz_corpus as K
z = B.Zstd(); plug = B.Plugin(); f = B.Front()
data = K.by_name("system", 5 * 131072 + 11, seed=3)
frames, st, fs = f.frames(data, 131072, 1, 3)
assert b"".join(z.decompress(fr, 131072) for fr in frames) == data
print("ok")
""" % os.path.join(B.ROOT, "tools")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, LD_PRELOAD=old[0]))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
