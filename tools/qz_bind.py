"""ctypes bindings shared by tests/, tools/ and bench.py.

* ``Zstd``     — libzstd >= 1.5.4 (the caller of the sequence producer and the judge of
                 its output).  Discovery order: $ZSTDLIB, the normally-built
                 1.5.7 inside pyarrow's libarrow.so through tools/zstdshim (round 4), a system
                 libzstd that exports ZSTD_registerSequenceProducer, the copy bundled in
                 pillow.libs (1.5.7 on the ROCm image, a ~4x slower build).
* ``Oracle``   — oracle/libqzstd_oracle.so (TEST INFRASTRUCTURE: only tests, smoke() and
                 bench.py's cpu_baseline leg may use it).
* ``Plugin``   — the product: libqatseqprod.so (drop-in C surface + qzstd_hip_* C-ABI).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "qat-zstd-plugin_amd")
PLUGIN_SO = os.environ.get("QZ_PLUGIN_SO") or os.path.join(PKG_DIR, "lib", "libqatseqprod.so")  # override: A/B kernel builds
ORACLE_SO = os.path.join(ROOT, "oracle", "libqzstd_oracle.so")

SEQ_ERROR = C.c_size_t(-1).value


class Sequence(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("litLength", C.c_uint32),
                ("matchLength", C.c_uint32), ("rep", C.c_uint32)]


PRODUCER_F = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(Sequence), C.c_size_t, C.c_void_p,
                         C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t)

# zstd parameter ids (include/qzstd_zstd_abi.h)
c_compressionLevel = 100
c_windowLog = 101
c_enableLongDistanceMatching = 160
c_checksumFlag = 201
c_nbWorkers = 400
c_stableInBuffer = 1006
c_validateSequences = 1009
c_enableSeqProducerFallback = 1014
c_maxBlockSize = 1015
c_searchForExternalRepcodes = 1016
c_blockSplitterLevel = 1017
ps_auto, ps_enable, ps_disable = 0, 1, 2
e_continue, e_flush, e_end = 0, 1, 2


SHIM_DIR = os.path.join(ROOT, "tools", "zstdshim")
SHIM_SO = os.path.join(SHIM_DIR, "libzstd-arrow.so")
_libzstd_path = None


def fast_libzstd() -> str | None:
    """tools/zstdshim: the normally-built zstd 1.5.7 inside pyarrow's libarrow.so behind the public ZSTD_* names (zstdshim.c says why: the
    only exported libzstd >= 1.5.4 on the image, Pillow's, is a 4x slower build).  Builds the shim when it is missing or older than its
    sources; returns its path when it resolved every name (zstdshim_ok), else None.  $QZ_ZSTD_NO_SHIM=1 turns it off."""
    if os.environ.get("QZ_ZSTD_NO_SHIM", "0") not in ("", "0"):
        return None
    srcs = [os.path.join(SHIM_DIR, "zstdshim.c"), os.path.join(SHIM_DIR, "zstd_names.h")]
    try:
        if not all(os.path.isfile(x) for x in srcs):
            return None
        if not os.path.isfile(SHIM_SO) or os.path.getmtime(SHIM_SO) < max(os.path.getmtime(x) for x in srcs):
            import subprocess
            tmp = "%s.%d.tmp" % (SHIM_SO, os.getpid())  # (several pytest workers may get here together: build aside, rename)
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", tmp, srcs[0], "-ldl", "-Wl,-soname,libzstd-arrow.so"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(tmp, SHIM_SO)
        lib = C.CDLL(SHIM_SO)
        return SHIM_SO if lib.zstdshim_ok() == 1 and hasattr(lib, "ZSTD_registerSequenceProducer") else None
    except Exception:  # noqa: BLE001 - no gcc, no libarrow.so, an image without pyarrow: the Pillow copy it is
        return None


def slow_libzstd() -> str:
    """an EXPORTED libzstd >= 1.5.4: $ZSTDLIB, the system's, the copy bundled in pillow.libs (a slow build: see fast_libzstd)"""
    cands = []
    env = os.environ.get("ZSTDLIB")
    if env:
        cands += [env] if os.path.isfile(env) else sorted(glob.glob(os.path.join(env, "libzstd.so*")))
    sysl = ctypes.util.find_library("zstd")
    if sysl:
        cands.append(sysl)
    cands += sorted(glob.glob("/usr/local/lib/libzstd.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libzstd.so*"))
    cands += sorted(glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd-*.so*"))
    cands += sorted(glob.glob("/usr/lib/python3/dist-packages/pillow.libs/libzstd-*.so*"))
    for c in cands:
        try:
            lib = C.CDLL(c)
            if hasattr(lib, "ZSTD_registerSequenceProducer"):
                # ctypes.util.find_library() gives a bare soname ("libzstd.so.1"): the Makefiles pass the result to gcc as a FILE and derive an
                # rpath from its directory, so what leaves here is always the absolute path of the file the loader mapped (round-5 ADVICE, medium)
                return c if os.path.isabs(c) else _mapped_path(c)
        except OSError:
            continue
    raise OSError("no libzstd >= 1.5.4 (ZSTD_registerSequenceProducer) found; set $ZSTDLIB")


def _mapped_path(soname: str) -> str:
    """absolute path of the already-dlopen()ed library the loader resolved `soname` to (from /proc/self/maps); raises OSError when it
    cannot be told — a bare soname must never reach a link line as a file name"""
    base = os.path.basename(soname)
    stem = base.split(".so")[0]
    try:
        with open("/proc/self/maps") as f:
            paths = {ln.split(None, 5)[5].strip() for ln in f if ln.count("/") and len(ln.split(None, 5)) == 6}
    except OSError:
        paths = set()
    hits = sorted(p for p in paths if os.path.basename(p) == base) or \
        sorted(p for p in paths if os.path.basename(p).startswith(stem + ".so") and os.path.realpath(p) in {os.path.realpath(q) for q in paths})
    for p in hits:
        if os.path.isfile(p):
            return p
    raise OSError("cannot resolve %s to a path" % soname)


def find_libzstd() -> str:
    """the libzstd every test, tool and bench leg runs: $ZSTDLIB when set, else the shim over libarrow.so's copy when it works, else slow_libzstd()"""
    global _libzstd_path
    if _libzstd_path is None:
        _libzstd_path = (None if os.environ.get("ZSTDLIB") else fast_libzstd()) or slow_libzstd()
    return _libzstd_path


def libzstd_build(path: str) -> str:
    """what a bench line says about the libzstd it ran"""
    if os.path.abspath(path) == os.path.abspath(SHIM_SO):
        lib = C.CDLL(path)
        lib.zstdshim_source.restype = C.c_char_p
        return "the copy inside %s through tools/zstdshim (a normal build)" % lib.zstdshim_source().decode()
    return path + (" (the image's Pillow copy: a ~4x slower build)" if "pillow.libs" in path else "")


class ZBounds(C.Structure):
    _fields_ = [("error", C.c_size_t), ("lowerBound", C.c_int), ("upperBound", C.c_int)]


class Zstd:
    def __init__(self, path: str | None = None):
        self.path = path or find_libzstd()
        L = self.lib = C.CDLL(self.path, mode=C.RTLD_GLOBAL)
        L.ZSTD_versionString.restype = C.c_char_p
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_CCtx_setParameter.restype = C.c_size_t
        L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_compress2.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_decompress.restype = C.c_size_t
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_getErrorName.argtypes = [C.c_size_t]
        L.ZSTD_getErrorName.restype = C.c_char_p
        self.has_producer_api = hasattr(L, "ZSTD_registerSequenceProducer")
        if not self.has_producer_api:  # libzstd 1.4.x: software baseline timing only
            return
        L.ZSTD_sequenceBound.argtypes = [C.c_size_t]
        L.ZSTD_sequenceBound.restype = C.c_size_t
        L.ZSTD_registerSequenceProducer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ZSTD_registerSequenceProducer.restype = None
        L.ZSTD_cParam_getBounds.argtypes = [C.c_int]
        L.ZSTD_cParam_getBounds.restype = ZBounds
        L.ZSTD_generateSequences.argtypes = [C.c_void_p, C.POINTER(Sequence), C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_generateSequences.restype = C.c_size_t
        L.ZSTD_CCtx_reset.argtypes = [C.c_void_p, C.c_int]
        L.ZSTD_CCtx_reset.restype = C.c_size_t

    def version(self) -> str:
        return self.lib.ZSTD_versionString().decode()

    def is_error(self, code: int) -> bool:
        return bool(self.lib.ZSTD_isError(code))

    def err(self, code: int) -> str:
        return self.lib.ZSTD_getErrorName(code).decode()

    def cctx(self, level: int, producer=None, state=None, fallback: bool = False,
             validate: bool = True, ext_repcodes: int | None = None, **params):
        """Create a CCtx; `producer` is an address/ctypes function pointer or None."""
        L = self.lib
        zc = L.ZSTD_createCCtx()
        assert zc
        self.set(zc, c_compressionLevel, level)
        if producer is not None:
            addr = C.cast(producer, C.c_void_p)
            L.ZSTD_registerSequenceProducer(zc, state, addr)
            self.set(zc, c_enableSeqProducerFallback, 1 if fallback else 0)
            if validate:
                self.set(zc, c_validateSequences, 1)
        if ext_repcodes is not None:
            self.set(zc, c_searchForExternalRepcodes, ext_repcodes)
        for k, v in params.items():
            self.set(zc, globals()["c_" + k], v)
        return zc

    def set(self, zc, pid: int, val: int):
        r = self.lib.ZSTD_CCtx_setParameter(zc, pid, val)
        if self.is_error(r):
            raise RuntimeError("ZSTD_CCtx_setParameter(%d,%d): %s" % (pid, val, self.err(r)))
        return r

    def free(self, zc):
        self.lib.ZSTD_freeCCtx(zc)

    def compress2(self, zc, data: bytes | memoryview, dst=None) -> bytes:
        n = len(data)
        cap = self.lib.ZSTD_compressBound(n)
        dst = dst if dst is not None and len(dst) >= cap else C.create_string_buffer(cap)
        src = (C.c_char * n).from_buffer_copy(data) if n else C.create_string_buffer(1)
        r = self.lib.ZSTD_compress2(zc, dst, cap, src, n)
        if self.is_error(r):
            raise RuntimeError("ZSTD_compress2: " + self.err(r))
        return dst.raw[:r]

    def compress_chunks(self, zc, data: bytes, chunk: int) -> tuple[int, list[bytes]]:
        """benchmark.c framing (reference test/benchmark.c:300-321): one frame per chunk."""
        total = 0
        frames = []
        cap = self.lib.ZSTD_compressBound(chunk)
        dst = C.create_string_buffer(cap)
        for o in range(0, len(data), chunk):
            f = self.compress2(zc, data[o:o + chunk], dst)
            total += len(f)
            frames.append(f)
        return total, frames

    def decompress(self, frame: bytes, size: int) -> bytes:
        dst = C.create_string_buffer(max(size, 1))
        r = self.lib.ZSTD_decompress(dst, size, frame, len(frame))
        if self.is_error(r):
            raise RuntimeError("ZSTD_decompress: " + self.err(r))
        return dst.raw[:r]


class OracleProfile(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("tableSize", "tileLog", "capLen", "minMatch", "farLog1",
                                          "farLog2", "lazy", "backExt", "nearTab", "window",
                                          "hashBytes", "extLog", "longSize", "repWin", "chainDepth", "subTileLog", "segLog")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Oracle:
    """CPU oracle — checker only (see oracle/qzstd_oracle.h)."""

    def __init__(self, path: str = ORACLE_SO):
        L = self.lib = C.CDLL(path)
        L.qzo_profile_for_level.argtypes = [C.c_int, C.c_size_t, C.POINTER(OracleProfile)]
        L.qzo_find_sequences.argtypes = [C.POINTER(OracleProfile), C.c_void_p, C.c_size_t,
                                         C.POINTER(Sequence), C.c_size_t]
        L.qzo_find_sequences.restype = C.c_size_t
        L.qzo_find_sequences_from.argtypes = [C.POINTER(OracleProfile), C.c_void_p, C.c_size_t, C.c_size_t,
                                              C.POINTER(Sequence), C.c_size_t]
        L.qzo_find_sequences_from.restype = C.c_size_t
        L.qzo_validate.argtypes = [C.POINTER(Sequence), C.c_size_t, C.c_size_t, C.c_size_t]
        L.qzo_reconstruct_check.argtypes = [C.POINTER(Sequence), C.c_size_t, C.c_void_p, C.c_size_t]
        L.qzo_reconstruct_check.restype = C.c_size_t
        L.qzo_seq_stats.argtypes = [C.POINTER(Sequence), C.c_size_t, C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.qzo_seq_stats.restype = None
        L.qzo_lz4s_decode.argtypes = [C.POINTER(Sequence), C.c_size_t, C.c_void_p, C.c_size_t]
        L.qzo_lz4s_decode.restype = C.c_size_t
        L.qzo_lz4s_encode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Sequence), C.c_size_t,
                                      C.c_void_p, C.c_size_t]
        L.qzo_lz4s_encode.restype = C.c_size_t
        self.producer_addr = C.cast(L.qzo_sequence_producer, C.c_void_p)

    def profile(self, level: int, block: int) -> OracleProfile:
        p = OracleProfile()
        if self.lib.qzo_profile_for_level(level, block, C.byref(p)) != 0:
            raise ValueError("bad level %d" % level)
        return p

    def find(self, prof: OracleProfile, data: bytes, cap: int | None = None, parse_from: int = 0):
        """sequences of a block; parse_from != 0: of the segment that starts there (data = the block up to the segment's end)"""
        n = len(data)
        cap = cap or (n // 3 + 1 + n // 1024 + 1)
        out = (Sequence * cap)()
        r = self.lib.qzo_find_sequences_from(C.byref(prof), data, n, parse_from, out, cap)
        return r, out

    def stats(self, seqs, n):
        a, b, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.qzo_seq_stats(seqs, n, C.byref(a), C.byref(b), C.byref(h))
        return a.value, b.value, h.value


def sequence_bound(n: int) -> int:
    """ZSTD_sequenceBound (zstd 1.5.x): n/3 + 1 + n/1024 + 1."""
    return n // 3 + 1 + n // 1024 + 1


class HipProfile(OracleProfile):
    """qzstd_hip_profile_t — same 12 x u32 layout as the oracle's profile."""


class HipBlock(C.Structure):
    _fields_ = [("srcOff", C.c_uint64), ("seqOff", C.c_uint64), ("srcLen", C.c_uint32), ("seqCap", C.c_uint32),
                ("parseFrom", C.c_uint32), ("mark", C.c_uint32)]


NSEQ_ERROR = 0xFFFFFFFF
MARK_COMPACT = 0x80000000  # qzstd_hip.h: QZSTD_HIP_MARK_COMPACT
NSEQ_REJECTED = 0xFFFFFFFE


class SvcReq(C.Structure):
    """qzstd_hip_svc_req_t (include/qzstd_hip.h): one block for the resident service"""
    _fields_ = [("hSrc", C.c_void_p), ("dSrc", C.c_void_p), ("hSeqs", C.c_void_p), ("hCount", C.c_void_p),
                ("srcLen", C.c_uint32), ("itemBytes", C.c_uint32), ("nItems", C.c_uint32), ("seqCapPerItem", C.c_uint32),
                ("slot", C.c_uint32), ("epoch", C.c_uint32), ("dWork", C.c_void_p)]


# every symbol include/qatseqprod.h and include/qzstd_hip.h declare
PLUGIN_SYMBOLS = [
    "QZSTD_version", "qatSequenceProducer", "QZSTD_startQatDevice", "QZSTD_stopQatDevice",
    "QZSTD_createSeqProdState", "QZSTD_freeSeqProdState", "QZSTD_hintSource", "QZSTD_hintSourceEx", "QZSTD_dropHints", "QZSTD_hintBroken", "QZSTD_hintStats", "QZSTD_failStats", "QZSTD_deviceStats",
    "qzstd_hip_last_error", "qzstd_hip_profile_for_level", "qzstd_hip_sequence_bound", "qzstd_hip_lds_bytes",
    "qzstd_hip_device_count", "qzstd_hip_device_name", "qzstd_hip_malloc", "qzstd_hip_free",
    "qzstd_hip_host_alloc", "qzstd_hip_host_free", "qzstd_hip_stream_create", "qzstd_hip_stream_destroy",
    "qzstd_hip_stream_sync", "qzstd_hip_stream_query", "qzstd_hip_stream_wait", "qzstd_hip_memcpy_h2d", "qzstd_hip_copy_in", "qzstd_hip_memcpy_d2h",
    "qzstd_hip_memset", "qzstd_hip_memcpy2d_d2h", "qzstd_hip_host_device_ptr", "qzstd_hip_workspace_bytes", "qzstd_hip_find_sequences",
    "qzstd_hip_service_submit", "qzstd_hip_service_stop", "qzstd_hip_service_poke", "qzstd_hip_service_progressive", "qzstd_hip_service_mark_broken", "qzstd_hip_service_info", "qzstd_hip_service_debug",
    "qzstd_hip_host_alloc_coherent", "qzstd_hip_device_numa_node", "qzstd_hip_host_alloc_on_node", "qzstd_hip_host_node_of", "qzstd_hip_occupancy",
]


class Plugin:
    """The product library, reached only through its C ABI (include/*.h)."""

    def __init__(self, path: str = PLUGIN_SO):
        if not os.path.isfile(path):
            raise OSError("%s missing: run `make -C qat-zstd-plugin_amd` (or __graft_entry__.build())" % path)
        L = self.lib = C.CDLL(path)
        L.QZSTD_version.restype = C.c_char_p
        L.QZSTD_createSeqProdState.restype = C.c_void_p
        L.QZSTD_freeSeqProdState.argtypes = [C.c_void_p]
        L.QZSTD_hintSource.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        if hasattr(L, "QZSTD_hintSourceEx"):
            L.QZSTD_hintSourceEx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]
            L.QZSTD_dropHints.argtypes = [C.c_void_p]
            L.QZSTD_dropHints.restype = None
        L.QZSTD_hintStats.argtypes = [C.c_void_p, C.POINTER(C.c_ulong * 4)]
        L.QZSTD_hintStats.restype = None
        L.QZSTD_failStats.argtypes = [C.c_void_p, C.POINTER(C.c_ulong * 8)]
        L.QZSTD_failStats.restype = None
        L.qatSequenceProducer.restype = C.c_size_t
        L.qatSequenceProducer.argtypes = [C.c_void_p, C.POINTER(Sequence), C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.c_int, C.c_size_t]
        L.qzstd_hip_last_error.restype = C.c_char_p
        L.qzstd_hip_profile_for_level.argtypes = [C.c_int, C.c_size_t, C.POINTER(HipProfile)]
        L.qzstd_hip_sequence_bound.argtypes = [C.c_size_t]
        L.qzstd_hip_sequence_bound.restype = C.c_size_t
        L.qzstd_hip_lds_bytes.argtypes = [C.c_int, C.c_uint32]
        L.qzstd_hip_lds_bytes.restype = C.c_size_t
        L.qzstd_hip_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.qzstd_hip_malloc.argtypes = [C.c_int, C.c_size_t]
        L.qzstd_hip_malloc.restype = C.c_void_p
        L.qzstd_hip_free.argtypes = [C.c_int, C.c_void_p]
        L.qzstd_hip_free.restype = None
        L.qzstd_hip_host_alloc.argtypes = [C.c_size_t]
        L.qzstd_hip_host_alloc.restype = C.c_void_p
        L.qzstd_hip_host_free.argtypes = [C.c_void_p]
        L.qzstd_hip_host_free.restype = None
        L.qzstd_hip_stream_create.argtypes = [C.c_int]
        L.qzstd_hip_stream_create.restype = C.c_void_p
        L.qzstd_hip_stream_destroy.argtypes = [C.c_int, C.c_void_p]
        L.qzstd_hip_stream_destroy.restype = None
        L.qzstd_hip_stream_sync.argtypes = [C.c_int, C.c_void_p]
        L.qzstd_hip_stream_query.argtypes = [C.c_int, C.c_void_p]
        L.qzstd_hip_memcpy_h2d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.qzstd_hip_memcpy_d2h.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.qzstd_hip_memset.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
        L.qzstd_hip_find_sequences.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32,
                                               C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.qzstd_hip_workspace_bytes.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.qzstd_hip_workspace_bytes.restype = C.c_size_t
        L.qzstd_hip_host_alloc_coherent.restype = C.c_void_p
        L.qzstd_hip_host_alloc_coherent.argtypes = [C.c_size_t]
        L.qzstd_hip_service_submit.argtypes = [C.c_int, C.c_int, C.POINTER(SvcReq)]
        L.qzstd_hip_service_info.argtypes = [C.c_int, C.POINTER(C.c_ulong * 8)]
        L.qzstd_hip_service_debug.argtypes = [C.c_int, C.POINTER(C.c_ulong * 8)]
        self.producer_addr = C.cast(L.qatSequenceProducer, C.c_void_p)

    def err(self) -> str:
        return self.lib.qzstd_hip_last_error().decode()

    def profile(self, level: int, block: int) -> HipProfile:
        p = HipProfile()
        if self.lib.qzstd_hip_profile_for_level(level, block, C.byref(p)) != 0:
            raise ValueError("bad level %d" % level)
        return p

    def check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.err()))

    def service_lane(self, slot: int, device: int = 0):
        """buffers of one caller of the resident service (what a slot of the plugin holds): returns an object with
        .run(block, level, item_bytes, timeout_s) -> (counts, Sequence array, cap per item) or None when not served"""
        return ServiceLane(self, slot, device)

    def find_batch(self, blocks: list[bytes], level: int = 1, device: int = 0, stride: int | None = None,
                   caps: list[int] | None = None, parse_from: list[int] | None = None, packed_tag: int = 0, launch_max_len: int | None = None):
        """Run the HIP match-finder over `blocks` through the C ABI (device memory managed
        with qzstd_hip_malloc / memcpy).  Returns (counts, list of Sequence arrays).
        packed_tag != 0: the items ask for PACKED entries (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT | tag, 8 bytes each, seqOff in 16-byte units);
        they are unpacked here into the same Sequence array (rep = the tag found), so callers compare them like the 16-byte ones."""
        L = self.lib
        nb = len(blocks)
        maxlen = max([len(b) for b in blocks] + [1])
        stride = stride or sequence_bound(maxlen)
        if launch_max_len is not None:  # (tests: a launch that understates its longest block)
            maxlen = launch_max_len
        offs, total = [], 0
        for b in blocks:
            offs.append(total)
            total += (len(b) + 15) & ~15
        total = max(total, 16)
        host_src = bytearray(total)
        for o, b in zip(offs, blocks):
            host_src[o:o + len(b)] = b
        desc = (HipBlock * nb)()
        for i, b in enumerate(blocks):
            desc[i].srcOff = offs[i]
            desc[i].seqOff = i * stride if not packed_tag else i * ((stride + 1) // 2)
            desc[i].mark = (MARK_COMPACT | (packed_tag & 0xFFF)) if packed_tag else 0
            desc[i].srcLen = len(b)
            desc[i].seqCap = caps[i] if caps else stride
            desc[i].parseFrom = parse_from[i] if parse_from else 0  # segment mode: the item parses [parseFrom, len) only
        d_src = L.qzstd_hip_malloc(device, total)
        d_desc = L.qzstd_hip_malloc(device, C.sizeof(desc))
        d_seqs = L.qzstd_hip_malloc(device, nb * stride * 16)
        d_cnt = L.qzstd_hip_malloc(device, nb * 4)
        work = L.qzstd_hip_workspace_bytes(level, nb, maxlen)
        d_work = L.qzstd_hip_malloc(device, work) if work else None
        try:
            if not (d_src and d_desc and d_seqs and d_cnt and (d_work or not work)):
                raise RuntimeError("qzstd_hip_malloc: " + self.err())
            src_buf = (C.c_char * total).from_buffer(host_src)
            self.check(L.qzstd_hip_memcpy_h2d(device, None, d_src, src_buf, total), "h2d src")
            self.check(L.qzstd_hip_memcpy_h2d(device, None, d_desc, desc, C.sizeof(desc)), "h2d desc")
            self.check(L.qzstd_hip_memset(device, None, d_cnt, 0, nb * 4), "memset")
            self.check(L.qzstd_hip_find_sequences(device, None, level, d_src, d_desc, nb, maxlen, d_seqs, d_cnt, d_work, work),
                       "qzstd_hip_find_sequences")
            cnt = (C.c_uint32 * nb)()
            seqs = (Sequence * (nb * stride))()
            self.check(L.qzstd_hip_memcpy_d2h(device, None, cnt, d_cnt, nb * 4), "d2h counts")
            self.check(L.qzstd_hip_memcpy_d2h(device, None, seqs, d_seqs, nb * stride * 16), "d2h seqs")
            self.check(L.qzstd_hip_stream_sync(device, None), "sync")
            if packed_tag:
                import numpy as np
                raw = np.frombuffer(seqs, dtype=np.uint64)  # the device buffer's first bytes: block i's packed entries start at 16 * seqOff
                out = np.zeros((nb * stride, 4), dtype=np.uint32)
                for i in range(nb):
                    n = cnt[i] if cnt[i] != NSEQ_ERROR else 0
                    v = raw[i * ((stride + 1) // 2) * 2:][:n]
                    out[i * stride:i * stride + n, 0] = (v & 0x1FFFF).astype(np.uint32)
                    out[i * stride:i * stride + n, 1] = ((v >> np.uint64(17)) & np.uint64(0x3FFFF)).astype(np.uint32)
                    out[i * stride:i * stride + n, 2] = ((v >> np.uint64(35)) & np.uint64(0x1FFFF)).astype(np.uint32)
                    out[i * stride:i * stride + n, 3] = (v >> np.uint64(52)).astype(np.uint32)
                seqs = (Sequence * (nb * stride)).from_buffer_copy(out.tobytes())
        finally:
            for p in (d_src, d_desc, d_seqs, d_cnt, d_work):
                if p:
                    L.qzstd_hip_free(device, p)
        return list(cnt), seqs, stride


class ServiceLane:
    """One request at a time through qzstd_hip_service_submit, by hand (tests, tools)."""
    MAX_ITEMS, ITEM_CAP, BLOCK_MAX = 32, 1371, 1 << 17

    def __init__(self, plug: Plugin, slot: int, device: int = 0):
        L = self.L = plug.lib
        self.plug, self.slot, self.device, self.epoch = plug, slot, device, 0
        self.hsrc = L.qzstd_hip_host_alloc_coherent(self.BLOCK_MAX + 64)
        self.hseq = L.qzstd_hip_host_alloc_coherent(self.MAX_ITEMS * self.ITEM_CAP * 16)
        self.hcnt = L.qzstd_hip_host_alloc_coherent(self.MAX_ITEMS * 4)
        self.dsrc = L.qzstd_hip_malloc(device, self.BLOCK_MAX + 64)
        self.dwork = None
        assert self.hsrc and self.hseq and self.hcnt and self.dsrc, plug.err()
        self.cnt = (C.c_uint32 * self.MAX_ITEMS).from_address(self.hcnt)
        self.seqs = (Sequence * (self.MAX_ITEMS * self.ITEM_CAP)).from_address(self.hseq)

    def run(self, block: bytes, level: int, item_bytes: int = 4096, timeout_s: float = 5.0):
        import time
        n = len(block)
        while -(-n // item_bytes) > self.MAX_ITEMS:
            item_bytes *= 2
        nit = -(-n // item_bytes)
        cap = self.MAX_ITEMS * self.ITEM_CAP // nit
        C.memmove(self.hsrc, block + bytes(16), n + 16)
        for k in range(nit):
            self.cnt[k] = 0
        self.epoch = self.epoch % 0xFFFFFF + 1
        work = None
        if self.plug.profile(level, self.BLOCK_MAX).chainDepth:  # a chain level: the request carries a chain scratch
            if not self.dwork:
                self.dwork = self.L.qzstd_hip_malloc(self.device, self.BLOCK_MAX * (8 * 4 + 4) + 32 * 5888 * 4)  # QZSTD_HIP_SVC_WORK_BYTES: one scratch per request
                assert self.dwork, self.plug.err()
            work = self.dwork
        rq = SvcReq(self.hsrc, self.dsrc, self.hseq, self.hcnt, n, item_bytes, nit, cap, self.slot, self.epoch, work)
        rc = self.L.qzstd_hip_service_submit(self.device, level, C.byref(rq))
        if rc == 1:
            return None
        assert rc == 0, self.plug.err()
        t0 = time.perf_counter()
        while not all(self.cnt[k] for k in range(nit)):
            assert time.perf_counter() - t0 < timeout_s, "service request timed out: counts %s" % [self.cnt[k] for k in range(nit)]
        counts = [self.cnt[k] for k in range(nit)]
        # the counts say how many entries; every entry carries the epoch in its fourth word when it has arrived (include/qzstd_hip.h)
        import numpy as np
        words = np.frombuffer((C.c_uint32 * (self.MAX_ITEMS * self.ITEM_CAP * 4)).from_address(self.hseq), dtype=np.uint32)
        for k, n in enumerate(counts):
            if n in (NSEQ_ERROR, NSEQ_REJECTED) or n > cap:
                continue
            while not (words[k * cap * 4 + 3:(k * cap + n) * 4:4] == self.epoch).all():
                assert time.perf_counter() - t0 < timeout_s, "entries of item %d never arrived" % k
        return counts, self.seqs, cap, item_bytes

    def close(self):
        L = self.L
        L.qzstd_hip_host_free(self.hsrc); L.qzstd_hip_host_free(self.hseq); L.qzstd_hip_host_free(self.hcnt)
        L.qzstd_hip_free(self.device, self.dsrc)
        if self.dwork:
            L.qzstd_hip_free(self.device, self.dwork)


FRONT_SO = os.path.join(PKG_DIR, "lib", "libqzstdfront.so")


class FrontParams(C.Structure):
    _fields_ = [("nThreads", C.c_int), ("level", C.c_int), ("chunkSize", C.c_size_t), ("segmentBytes", C.c_size_t),
                ("extRepcodes", C.c_int), ("useProducer", C.c_int)]


class Front:
    """include/qzstd_frontend.h: the batch front-end for the host entropy stage (a pool of CCtx threads, every segment announced
    one claim ahead).  Load libzstd (Zstd()) and the plugin first: the library links against both."""

    def __init__(self, path: str = FRONT_SO):
        if not os.path.isfile(path):
            raise OSError("%s missing: `make -C qat-zstd-plugin_amd front` needs a libzstd >= 1.5.4" % path)
        # The library's ZSTD_* references bind through the process's global scope first: normally that is the libzstd Zstd() loaded
        # (RTLD_GLOBAL).  Under a tool that preloads libraries (rocprofv3: libdw -> the system's libzstd 1.4.8) ANOTHER libzstd sits in
        # front of it, and the front-end's ZSTD_CCtx_setParameter(ZSTD_c_enableSeqProducerFallback) would go to a library that does not
        # know the parameter: then the library is bound to its own dependencies first (RTLD_DEEPBIND: the libzstd it was linked against).
        mode = C.DEFAULT_MODE
        try:
            want = C.cast(Zstd().lib.ZSTD_versionNumber, C.c_void_p).value
            seen = C.cast(C.CDLL(None).ZSTD_versionNumber, C.c_void_p).value
            if seen != want:
                mode |= os.RTLD_DEEPBIND
        except (AttributeError, OSError):
            pass
        F = self.lib = C.CDLL(path, mode=mode)
        F.QZSTD_createFront.restype = C.c_void_p
        F.QZSTD_createFront.argtypes = [C.POINTER(FrontParams)]
        F.QZSTD_frontFrameStride.restype = C.c_size_t
        F.QZSTD_frontFrameStride.argtypes = [C.c_void_p]
        F.QZSTD_frontCompress.restype = C.c_size_t
        F.QZSTD_frontCompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        F.QZSTD_frontCompact.restype = C.c_size_t
        F.QZSTD_frontCompact.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t]
        F.QZSTD_frontStats.argtypes = [C.c_void_p, C.POINTER(C.c_ulong)]
        F.QZSTD_frontFailStats.argtypes = [C.c_void_p, C.POINTER(C.c_ulong)]
        F.QZSTD_freeFront.argtypes = [C.c_void_p]

    def frames(self, data: bytes, chunk: int, level: int, threads: int, segment: int = 0, jobs: int = 1, ext_rep: int = 0, each=None):
        """-> (frames of the last pass, [announced, per block] blocks, fail stats[8]) of `jobs` passes of `data` through one front;
        each(job, frames) is called after every pass"""
        F = self.lib
        prm = FrontParams(threads, level, chunk, segment, ext_rep, 1)
        f = F.QZSTD_createFront(C.byref(prm))
        if not f:
            raise RuntimeError("QZSTD_createFront failed")
        try:
            stride = F.QZSTD_frontFrameStride(f)
            n = (len(data) + chunk - 1) // chunk
            dst = C.create_string_buffer(n * stride)
            sizes = (C.c_size_t * n)()
            for j in range(jobs):
                got = F.QZSTD_frontCompress(f, data, len(data), dst, len(dst), sizes)
                if got != n:
                    raise RuntimeError("QZSTD_frontCompress returned %d, expected %d frames" % (got, n))
                if each is not None:
                    raw = dst.raw
                    each(j, [raw[c * stride:c * stride + sizes[c]] for c in range(n)])
            st, fs = (C.c_ulong * 2)(), (C.c_ulong * 8)()
            F.QZSTD_frontStats(f, st)
            F.QZSTD_frontFailStats(f, fs)
            return [dst.raw[c * stride:c * stride + sizes[c]] for c in range(n)], list(st), list(fs)
        finally:
            F.QZSTD_freeFront(f)


if __name__ == "__main__":  # `python tools/qz_bind.py --libzstd`: the Makefiles' default ZSTDLIB
    import sys
    if "--libzstd-exported" in sys.argv:  # the product Makefile's default: an EXPORTED libzstd >= 1.5.4, never the shim
        print(slow_libzstd())
    elif "--libzstd" in sys.argv:  # what the tests, tools and bench legs run (the shim over libarrow.so's copy when it works)
        print(find_libzstd())
