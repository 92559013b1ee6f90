"""CPU (no GPU): host logic of the drop-in surface, the C ABI's symbol table, the level
tables, the zstd ABI slice, and BASELINE config #1 (plugin registered, no device ->
libzstd software fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import qz_bind as B
import qz_corpus as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def has_gpu(plugin):
    return plugin.lib.qzstd_hip_device_count() > 0


# ------------------------------------------------------------------ C ABI / headers
def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))  # drop macros
    names = set(re.findall(r"\b(QZSTD_[a-z]\w+|qatSequenceProducer|qzstd_hip_\w+)\s*\(", text))
    return {n for n in names if not n.endswith("_t")}


def test_library_exports_every_declared_symbol(plugin):
    """the C-ABI library loads and exports every function include/*.h declares"""
    declared = declared_functions("qatseqprod.h") | declared_functions("qzstd_hip.h")
    assert {"QZSTD_startQatDevice", "QZSTD_createSeqProdState", "qatSequenceProducer",
            "qzstd_hip_find_sequences"} <= declared
    assert declared == set(B.PLUGIN_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(plugin.lib, name), "libqatseqprod.so does not export " + name


def test_front_end_library_exports_what_its_header_declares(plugin):
    """include/qzstd_frontend.h <-> lib/libqzstdfront.so (built when a libzstd >= 1.5.4 is found; it is in this image)"""
    import ctypes
    declared = declared_functions("qzstd_frontend.h")
    assert declared == {"QZSTD_createFront", "QZSTD_frontFrameStride", "QZSTD_frontCompress", "QZSTD_frontCompact",
                        "QZSTD_frontStats", "QZSTD_frontFailStats", "QZSTD_freeFront"}
    B.Zstd()  # libzstd first (RTLD_GLOBAL): the front-end links against it
    lib = ctypes.CDLL(os.path.join(B.PKG_DIR, "lib", "libqzstdfront.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), "libqzstdfront.so does not export " + name


def test_static_library_has_the_reference_artefact_name():
    """libqatseqprod.a / .so / qatseqprod.h: artefact names of reference src/Makefile:85-95"""
    assert os.path.isfile(os.path.join(B.PKG_DIR, "lib", "libqatseqprod.a"))
    assert os.path.isfile(os.path.join(B.PKG_DIR, "lib", "libqatseqprod.so"))
    assert os.path.isfile(os.path.join(ROOT, "include", "qatseqprod.h"))
    out = subprocess.check_output(["nm", "-g", os.path.join(B.PKG_DIR, "lib", "libqatseqprod.a")], text=True)
    for sym in ("QZSTD_version", "QZSTD_startQatDevice", "QZSTD_stopQatDevice", "QZSTD_createSeqProdState",
                "QZSTD_freeSeqProdState", "qatSequenceProducer"):
        assert re.search(r" T %s$" % sym, out, re.M), sym


def test_product_does_not_link_or_name_the_oracle():
    """the product path must not route through oracle/ (test infrastructure)"""
    so = os.path.join(B.PKG_DIR, "lib", "libqatseqprod.so")
    needed = subprocess.check_output(["readelf", "-d", so], text=True)
    assert "oracle" not in needed
    syms = subprocess.check_output(["nm", "-D", so], text=True)
    assert "qzo_" not in syms
    for dirpath, _, files in os.walk(B.PKG_DIR):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "qzstd_oracle.h" not in txt and "libqzstd_oracle" not in txt, os.path.join(dirpath, f)


def test_version_matches_reference_macros(plugin):
    assert plugin.lib.QZSTD_version() == b"0.2.0"  # reference src/qatseqprod.h:50-55
    hdr = open(os.path.join(ROOT, "include", "qatseqprod.h")).read()
    assert '#define QZSTD_VERSION "0.2.0"' in hdr
    for name, val in (("QZSTD_OK", "0"), ("QZSTD_STARTED", "1"), ("QZSTD_FAIL", "-1"), ("QZSTD_UNSUPPORTED", "-2")):
        assert re.search(r"%s\s*=\s*%s\b" % (name, re.escape(val)), hdr), name


# ------------------------------------------------------------------ level tables / LDS budget
@pytest.mark.parametrize("level", list(range(1, 13)) + [0x100 | l for l in range(1, 13)])
@pytest.mark.parametrize("block", [1, 1000, 32768, 32769, 65536, 65537, 100000, 131072])
def test_profile_tables_agree(plugin, oracle, level, block):
    assert plugin.profile(level, block).as_dict() == oracle.profile(level, block).as_dict()


def test_workspace_per_level(plugin):
    """levels >= 5 keep their hash chains in device memory: per position of every work item a chain entry of four links (16 B)
    and a dense array of 4 B per position (the first links once more, for the history pass of segment items); below the chain
    levels only that array: the parse words of a launch, which parses after its tile loop"""
    W = plugin.lib.qzstd_hip_workspace_bytes
    for level in range(1, 5):
        assert plugin.profile(level, 131072).chainDepth == 0
        for lv in (level, level | 0x100):
            assert W(lv, 100, 131072) == 100 * (131072 * 4 + 131072 // 8) and W(lv, 3, 1000) == 3 * (1024 * 4 + 1024 // 8)  # words + the windows' start masks
        assert plugin.profile(level, 131072).subTileLog == (6 if level == 2 else 0)
    for level in range(5, 13):
        assert plugin.profile(level, 131072).chainDepth in (8, 12, 32, 40, 64) and plugin.profile(level, 131072).subTileLog == 6
        assert W(level, 100, 131072) == 100 * 131072 * 20
        assert W(level, 3, 1000) == 3 * 1024 * 20
    assert W(6, 1, 131073) == 0 and W(0, 1, 1000) == 0


def test_repcode_aware_parse_follows_libzstd_default(plugin):
    """libzstd resolves ZSTD_c_searchForExternalRepcodes = auto to "on" from level 10; below that the
    caller has to ask for it (level | QZSTD_HIP_LEVEL_REPCODES, env QZSTD_HIP_EXT_REPCODES=1)"""
    for level in range(1, 13):
        assert (plugin.profile(level, 131072).repWin != 0) == (level >= 10)
        assert plugin.profile(level | 0x100, 131072).repWin == 16


def test_profile_rejects_bad_levels(plugin, oracle):
    for lvl in (0, -1, 13, 22, 0x100, 0x10D, 0x201):
        assert plugin.lib.qzstd_hip_profile_for_level(lvl, 131072, C.byref(B.HipProfile())) != 0
        assert oracle.lib.qzo_profile_for_level(lvl, 131072, C.byref(B.OracleProfile())) != 0


def test_lds_budget_fits_gfx950(plugin):
    """160 KiB LDS per CU (MI355X_MICROARCH.md); every level / block size must fit one workgroup"""
    for level in range(1, 13):
        for blk in (1, 4096, 32768, 32769, 65536, 65537, 131072):
            need = plugin.lib.qzstd_hip_lds_bytes(level, blk)
            assert 0 < need <= 163840, (level, blk, need)
    assert plugin.lib.qzstd_hip_lds_bytes(1, 131073) == 0
    assert plugin.lib.qzstd_hip_lds_bytes(0, 1000) == 0
    # levels 1-2 (the headline config): two workgroups per CU; levels >= 3 trade that for bigger tables
    assert plugin.lib.qzstd_hip_lds_bytes(1, 131072) * 2 <= 163840
    assert plugin.lib.qzstd_hip_lds_bytes(2, 32768) * 2 <= 163840
    assert plugin.lib.qzstd_hip_lds_bytes(3, 131072) > 81920


def test_sequence_bound_matches_libzstd(plugin, zstd):
    for n in (0, 1, 3, 1023, 1024, 32768, 100001, 131072, 4 << 20):
        assert plugin.lib.qzstd_hip_sequence_bound(n) == zstd.lib.ZSTD_sequenceBound(n) == B.sequence_bound(n)


# ------------------------------------------------------------------ zstd ABI slice
def test_zstd_abi_header_values(zstd):
    """parameter ids / bounds declared in include/qzstd_zstd_abi.h, checked against the live library"""
    hdr = open(os.path.join(ROOT, "include", "qzstd_zstd_abi.h")).read()
    ids = {"compressionLevel": 100, "nbWorkers": 400, "enableLongDistanceMatching": 160}
    for name, val in ids.items():
        assert re.search(r"ZSTD_c_%s\s*=\s*%d\b" % (name, val), hdr)
    exp = {1006: "stableInBuffer", 1009: "validateSequences", 1014: "enableSeqProducerFallback",
           1015: "maxBlockSize", 1016: "searchForExternalRepcodes", 1017: "blockSplitterLevel"}
    for pid, name in exp.items():
        assert re.search(r"=\s*%d,?\s*/\* ZSTD_c_%s \*/" % (pid, name), hdr), name
    b = zstd.lib.ZSTD_cParam_getBounds(1014)
    assert (zstd.lib.ZSTD_isError(b.error), b.lowerBound, b.upperBound) == (0, 0, 1)
    b = zstd.lib.ZSTD_cParam_getBounds(1015)
    assert (b.lowerBound, b.upperBound) == (1024, 131072)
    b = zstd.lib.ZSTD_cParam_getBounds(1016)
    assert (b.lowerBound, b.upperBound) == (0, 2)
    b = zstd.lib.ZSTD_cParam_getBounds(1009)
    assert (b.lowerBound, b.upperBound) == (0, 1)
    assert C.sizeof(B.Sequence) == 16


def test_libzstd_rejects_unsupported_combinations(zstd, oracle):
    """limitations the reference documents (src/qatseqprod.h:96-108): nbWorkers > 0 and LDM"""
    data = K.text(1, 200000)
    for pid in (B.c_nbWorkers, B.c_enableLongDistanceMatching):
        zc = zstd.cctx(1, producer=oracle.producer_addr, state=None)
        try:
            zstd.set(zc, pid, 1)
        except RuntimeError:
            zstd.free(zc)
            continue  # a libzstd built without multithreading refuses nbWorkers outright
        with pytest.raises(RuntimeError):
            zstd.compress2(zc, data)
        zstd.free(zc)


# ------------------------------------------------------------------ config #1: no device -> fallback
def test_no_device_start_fails_and_is_idempotent(plugin):
    if has_gpu(plugin):
        pytest.skip("a GPU is visible: the no-device path cannot be exercised here")
    L = plugin.lib
    assert L.QZSTD_startQatDevice() == -1  # QZSTD_FAIL
    assert L.QZSTD_startQatDevice() == -1
    L.QZSTD_stopQatDevice()                # safe when never started (reference :428-449)
    L.QZSTD_stopQatDevice()


def test_state_lifecycle_null_safe(plugin):
    L = plugin.lib
    L.QZSTD_freeSeqProdState(None)
    st = L.QZSTD_createSeqProdState()
    assert st
    L.QZSTD_freeSeqProdState(st)


def test_producer_guards_without_device(plugin):
    """guards run before the device check (reference :1123-1137 then :1140)"""
    L = plugin.lib
    st = L.QZSTD_createSeqProdState()
    seqs = (B.Sequence * 64)()
    src = C.create_string_buffer(K.text(1, 100))
    for args in ((0, 1 << 17, None, 0), (13, 1 << 17, None, 0), (1, 50, None, 0), (1, 1 << 17, src, 0),
                 (1, 1 << 17, None, 7)):
        level, window, dct, dsz = args
        assert L.qatSequenceProducer(st, seqs, 64, src, 100, dct, dsz, level, window) == B.SEQ_ERROR
    L.QZSTD_freeSeqProdState(st)


def test_device_down_counts_and_retries_every_1000th_block(plugin):
    """reference :88, :1140-1152: each failed block bumps a counter; the 1000th re-probes the device"""
    if has_gpu(plugin):
        pytest.skip("needs the no-device condition")
    L = plugin.lib
    st = L.QZSTD_createSeqProdState()
    seqs = (B.Sequence * 64)()
    src = C.create_string_buffer(K.text(1, 100))
    for _ in range(2500):
        assert L.qatSequenceProducer(st, seqs, 64, src, 100, None, 0, 1, 1 << 17) == B.SEQ_ERROR
    L.QZSTD_freeSeqProdState(st)
    L.QZSTD_stopQatDevice()


@pytest.mark.parametrize("level", [1, 6, 12])
def test_config1_fallback_roundtrip_equals_software(plugin, zstd, level):
    """BASELINE config #1: one 128 KiB block, plugin registered, no device: with
    ZSTD_c_enableSeqProducerFallback the frame is byte-identical to plugin-unregistered zstd."""
    if has_gpu(plugin):
        pytest.skip("needs the no-device condition")
    data = K.text(1, 131072)
    plugin.lib.QZSTD_startQatDevice()
    st = plugin.lib.QZSTD_createSeqProdState()
    zc = zstd.cctx(level, producer=plugin.producer_addr, state=st, fallback=True)
    frame = zstd.compress2(zc, data)
    zstd.free(zc)
    zc = zstd.cctx(level)
    sw = zstd.compress2(zc, data)
    zstd.free(zc)
    assert frame == sw
    assert zstd.decompress(frame, len(data)) == data
    # without the fallback the same call must fail loudly (no silent CPU path inside the plugin)
    zc = zstd.cctx(level, producer=plugin.producer_addr, state=st, fallback=False)
    with pytest.raises(RuntimeError, match="sequence producer"):
        zstd.compress2(zc, data)
    zstd.free(zc)
    plugin.lib.QZSTD_freeSeqProdState(st)
    plugin.lib.QZSTD_stopQatDevice()


def test_hint_without_device_is_refused(plugin):
    if has_gpu(plugin):
        pytest.skip("needs the no-device condition")
    st = plugin.lib.QZSTD_createSeqProdState()
    buf = C.create_string_buffer(K.text(1, 300000))
    assert plugin.lib.QZSTD_hintSource(st, buf, 300000, 131072, 1) != 0
    assert plugin.lib.QZSTD_hintSource(st, buf, 300000, 131073, 1) != 0
    assert plugin.lib.QZSTD_hintSource(None, buf, 300000, 131072, 1) != 0
    plugin.lib.QZSTD_freeSeqProdState(st)


def test_hot_path_fails_loudly_without_gpu(plugin):
    """no CPU fallback inside the C ABI: the launch entry point reports an error"""
    if has_gpu(plugin):
        pytest.skip("needs the no-device condition")
    blk = B.HipBlock()
    dummy = C.create_string_buffer(64)
    rc = plugin.lib.qzstd_hip_find_sequences(0, None, 1, dummy, C.byref(blk), 1, 16, dummy, dummy, None, 0)
    assert rc != 0 and plugin.err()
    assert plugin.lib.qzstd_hip_malloc(0, 4096) is None


def test_c_roundtrip_program_config1(plugin, tmp_path):
    """the C counterpart of the reference's test/test.c builds against the drop-in header/library
    and passes (exit code 0) with the software fallback when no device is present"""
    zpath = B.find_libzstd()
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "ZSTDLIB=" + zpath], stdout=subprocess.DEVNULL)
    f = tmp_path / "dickens_like.bin"
    f.write_bytes(K.text(1, 131072))
    out = subprocess.run([os.path.join(tdir, "test"), str(f)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout and "Source size: 131072" in out.stdout
    bad = subprocess.run([os.path.join(tdir, "test"), str(tmp_path / "missing")], capture_output=True, text=True)
    assert bad.returncode != 0


def test_benchmark_tool_software_mode_and_no_device(plugin, tmp_path):
    """counterpart of the reference's test/benchmark.c: -m0 (software zstd) works anywhere; -m1 without a
    device fails because the tool, like the reference's (:261-267, :309-311), does not enable the fallback"""
    zpath = B.find_libzstd()
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "benchmark", "ZSTDLIB=" + zpath], stdout=subprocess.DEVNULL)
    f = tmp_path / "corpus.bin"
    f.write_bytes(K.mix(3, 5 * 65536 + 123))
    exe = os.path.join(tdir, "benchmark")
    out = subprocess.run([exe, "-m0", "-t2", "-l2", "-c64K", "-L3", "-E2", str(f)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stderr.count("PASS") == 2 and "Latency (us): P25" in out.stderr
    assert "software zstd level 3 chunk 65536 threads 2" in out.stderr
    if not has_gpu(plugin):
        bad = subprocess.run([exe, "-m1", "-c64K", str(f)], capture_output=True, text=True)
        assert bad.returncode != 0 and "Compress failed" in bad.stderr
    assert subprocess.run([exe, "-t0", str(f)], capture_output=True).returncode != 0      # option validation
    assert subprocess.run([exe, "-L13", str(f)], capture_output=True).returncode != 0


def test_fuzz_adapter_exports_the_five_hooks():
    """reference test/fuzzing/qatseqprodfuzzer.c:41-74: the symbols upstream zstd's fuzzers look for"""
    fdir = os.path.join(B.PKG_DIR, "test", "fuzzing")
    subprocess.check_call(["make", "-C", fdir], stdout=subprocess.DEVNULL)
    syms = subprocess.check_output(["nm", os.path.join(fdir, "qatseqprodfuzzer.o")], text=True)
    for s in ("FUZZ_seqProdSetup", "FUZZ_seqProdTearDown", "FUZZ_createSeqProdState", "FUZZ_freeSeqProdState",
              "FUZZ_thirdPartySeqProd"):
        assert re.search(r" T %s$" % s, syms, re.M), s
    assert " U qatSequenceProducer" in syms and " U QZSTD_stopQatDevice" not in syms


def test_bench_rank_sees_its_own_gpu_only(monkeypatch):
    """bench.py --gpus N: one process per GPU, and the plugin INSIDE rank r (front-end, slots, services) must use GPU r and no other —
    the rank narrows HIP_VISIBLE_DEVICES before HIP starts, indexing into whatever the job was started with"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("qz_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    assert bench.narrow_to_own_gpu(1, 0) is None and "HIP_VISIBLE_DEVICES" not in os.environ  # one rank: nothing is narrowed
    assert bench.narrow_to_own_gpu(8, 5) is None and os.environ["HIP_VISIBLE_DEVICES"] == "5"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert bench.narrow_to_own_gpu(4, 2) == "4,5,6,7" and os.environ["HIP_VISIBLE_DEVICES"] == "6"


def test_bench_ranks_share_out_the_cpus_of_their_gpus_node(monkeypatch):
    """bench.py --gpus N on a two-socket node: a rank's threads go to the CPUs of the NUMA node its GPU hangs off, the ranks of one node share
    them out in whole runs of the sibling-ordered list; unknown nodes, missing lists or too few CPUs leave the affinity alone.  The planning
    is a pure function (tools/qz_shard.py); the binding itself never costs the bench line (any exception -> not bound)"""
    import importlib.util
    import qz_shard as S
    assert S.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and S.parse_cpulist("") == [] and S.parse_cpulist("a-b") == []
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    allowed = list(range(256))
    got = [S.plan_rank_cpus(r, nodes, cpus, allowed) for r in range(8)]
    assert all(len(g) == 32 for g in got)
    assert sorted(c for g in got[:4] for c in g) == sorted(cpus[0]) and sorted(c for g in got[4:] for c in g) == sorted(cpus[1])  # a partition per node
    assert S.plan_rank_cpus(0, [0], {0: list(range(16))}, list(range(16))) == list(range(16))
    assert S.plan_rank_cpus(1, nodes, cpus, list(range(0, 256, 2))) == [c for c in cpus[0] if c % 2 == 0][16:32]  # only what the process may use
    assert S.plan_rank_cpus(0, [-1, -1], cpus, allowed) == [] and S.plan_rank_cpus(0, [0, 0], {1: [1, 2, 3, 4]}, allowed) == []
    assert S.plan_rank_cpus(0, [0, 0, 0, 0], {0: [0, 1, 2, 3, 4, 5, 6]}, allowed) == []  # fewer than two CPUs per rank: hands off
    assert S.plan_rank_cpus(9, nodes, cpus, allowed) == []

    spec = importlib.util.spec_from_file_location("qz_bench2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakePlug:
        class lib:
            @staticmethod
            def qzstd_hip_device_numa_node(d):
                return 0

    class FakeDist:
        @staticmethod
        def all_gather_object(out, obj):
            for i in range(len(out)):
                out[i] = obj

    before = os.sched_getaffinity(0)
    try:
        r = bench.bind_rank_to_its_gpus_node(FakePlug, 2, 1, FakeDist)
        assert isinstance(r, dict) and "bound" in r
        if r["bound"]:
            assert os.sched_getaffinity(0) <= before and len(os.sched_getaffinity(0)) == r["cpus"] >= 2
    finally:
        os.sched_setaffinity(0, before)

    class Broken:
        class lib:
            @staticmethod
            def qzstd_hip_device_numa_node(d):
                raise OSError("no such symbol")
    assert bench.bind_rank_to_its_gpus_node(Broken, 2, 0, FakeDist)["bound"] is False and os.sched_getaffinity(0) == before
