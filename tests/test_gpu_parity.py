"""GPU parity: the HIP match-finder (through the C ABI) must equal the CPU oracle
sequence-for-sequence, and its output must satisfy the consumer's rules."""
import ctypes as C

import numpy as np
import pytest

import qz_bind as B
import qz_corpus as K

pytestmark = pytest.mark.gpu


def seqs_to_np(seqs, start, n):
    a = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)
    return a[start:start + n, :3].copy()


def check_blocks(gpu_plugin, oracle, blocks, level=1, packed_tag=0):
    counts, seqs, stride = gpu_plugin.find_batch(blocks, level, packed_tag=packed_tag)
    for i, blk in enumerate(blocks):
        prof = oracle.profile(level, len(blk))
        want_n, want = oracle.find(prof, blk, cap=stride)
        assert counts[i] == (want_n if want_n != B.SEQ_ERROR else B.NSEQ_ERROR), \
            "block %d (len %d): count %d vs oracle %d" % (i, len(blk), counts[i], want_n)
        if want_n == B.SEQ_ERROR:
            continue
        got = seqs_to_np(seqs, i * stride, want_n)
        exp = seqs_to_np(want, 0, want_n)
        if not np.array_equal(got, exp):
            bad = int(np.nonzero((got != exp).any(axis=1))[0][0])
            raise AssertionError("block %d (len %d): first differing sequence %d: gpu %s oracle %s" % (
                i, len(blk), bad, got[bad], exp[bad]))
        if packed_tag:  # every packed entry, the delimiter included, carries the item's 12-bit tag
            tags = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)[i * stride:i * stride + want_n, 3]
            assert (tags == packed_tag).all(), "block %d: packed entries without the tag %#x" % (i, packed_tag)
            continue
        sub = (B.Sequence * want_n).from_buffer_copy(bytes(C.string_at(C.addressof(seqs) + i * stride * 16, want_n * 16)))
        assert oracle.lib.qzo_validate(sub, want_n, len(blk), 0) == 0
        assert oracle.lib.qzo_reconstruct_check(sub, want_n, blk, len(blk)) == 0


def test_single_text_block(gpu_plugin, oracle):
    check_blocks(gpu_plugin, oracle, [K.text(1, 131072)])


@pytest.mark.parametrize("gen", ["text", "binary", "weblog", "mixed_entropy", "random", "system"])
def test_corpora_128k(gpu_plugin, oracle, gen):
    data = K.by_name(gen, 8 * 131072, seed=7)
    check_blocks(gpu_plugin, oracle, [data[o:o + 131072] for o in range(0, len(data), 131072)])


def test_edge_sizes(gpu_plugin, oracle):
    base = K.text(3, 140000)
    sizes = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 1028, 1029, 2047, 2048, 2053,
             4096, 10000, 32767, 32768, 32769, 65535, 65536, 65537, 100001, 131071, 131072]
    check_blocks(gpu_plugin, oracle, [base[:s] for s in sizes])


@pytest.mark.parametrize("level", [1, 12, 0x102, 7])
def test_degenerate_blocks(gpu_plugin, oracle, level):
    blocks = [bytes(131072), b"\xff" * 70000, b"ab" * 50000, b"abc" * 40000, (b"0123456789" * 13108)[:131072],
              bytes(range(256)) * 512, b"x" + bytes(5000), K.incompressible(9, 131072),
              (b"abcdefgh" * 5 + b"X") * 3000, b"".join(b"record%05d;" % (i % 97) + bytes(53) for i in range(2000))]
    check_blocks(gpu_plugin, oracle, blocks, level)


@pytest.mark.parametrize("level", [1, 2, 3, 5, 6, 9, 10, 12, 0x101, 0x102, 0x103, 0x105, 0x106])  # 0x100 = QZSTD_HIP_LEVEL_REPCODES
def test_levels(gpu_plugin, oracle, level):
    data = K.mix(11, 6 * 131072)
    check_blocks(gpu_plugin, oracle, [data[o:o + 131072] for o in range(0, len(data), 131072)], level)


@pytest.mark.parametrize("level", [1, 3, 6, 12, 0x101])
def test_packed_entries_equal_the_oracle(gpu_plugin, oracle, level):
    """round 6: PACKED result entries (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT — offset 17 | litLength 18 | matchLength 17 | tag 12 bits in one
    8-byte store, what the announcements ask for so that the kernel pushes half the bytes over PCIe): the same sequences, entry for
    entry, as the 16-byte form and the oracle — on the blocks with the longest matches and literal runs there are (128 KiB of zeros: one
    match of 131 071; incompressible: one delimiter of 131 072 literals), ragged sizes, every kernel family"""
    data = K.mix(21, 3 * 131072)
    blocks = [data[o:o + 131072] for o in range(0, len(data), 131072)] + [bytes(131072), K.incompressible(5, 131072), b"ab" * 65536,
                                                                            K.text(2, 77777), K.text(4, 5), b"", K.weblog(3, 32768)]
    check_blocks(gpu_plugin, oracle, blocks, level, packed_tag=0xABC if level != 3 else 1)


def test_block_32k_and_64k(gpu_plugin, oracle):
    data = K.weblog(4, 16 * 32768)
    check_blocks(gpu_plugin, oracle, [data[o:o + 32768] for o in range(0, len(data), 32768)], 12)
    check_blocks(gpu_plugin, oracle, [data[o:o + 65536] for o in range(0, len(data), 65536)], 1)


def test_capacity_rule(gpu_plugin, oracle):
    """count >= cap-1 must be reported as an error (reference src/qatseqprod.c:1073-1076, :1318)."""
    blk = K.text(5, 131072)
    prof = oracle.profile(1, len(blk))
    n_full, _ = oracle.find(prof, blk)
    for cap in (n_full + 2, n_full + 1, n_full, 100, 3):
        counts, seqs, stride = gpu_plugin.find_batch([blk], 1, caps=[cap])
        want_n, _ = oracle.find(prof, blk, cap=cap)
        assert counts[0] == (want_n if want_n != B.SEQ_ERROR else B.NSEQ_ERROR), cap
    assert oracle.find(prof, blk, cap=n_full + 2)[0] == n_full
    assert oracle.find(prof, blk, cap=n_full + 1)[0] == B.SEQ_ERROR


@pytest.mark.parametrize("level,seed", [(1, 1), (1, 2), (3, 3), (0x101, 4), (12, 5), (0x104, 6), (6, 7), (0x108, 8)])
def test_randomised_blocks(gpu_plugin, oracle, level, seed):
    """ragged random batch: random sizes (incl. tiny and ring-wrapping ones), random content kinds,
    repeated / shifted copies that create far (> 40 KiB) and very long matches"""
    import random
    rng = random.Random(seed)
    pool = {k: K.by_name(k, 300000, seed=seed + i) for i, k in enumerate(["text", "binary", "weblog", "mixed_entropy", "random"])}
    blocks = []
    for _ in range(96):
        n = rng.choice([rng.randrange(0, 600), rng.randrange(600, 50000), rng.randrange(50000, 131073), 131072])
        kind = rng.choice(list(pool))
        o = rng.randrange(0, 300000 - 131072)
        b = bytearray(pool[kind][o:o + n])
        if n > 60000 and rng.random() < 0.5:  # plant a far, long repeat
            src = rng.randrange(0, n // 4)
            ln = rng.randrange(100, 20000)
            dst = rng.randrange(n // 2, max(n // 2 + 1, n - ln))
            b[dst:dst + ln] = b[src:src + ln][:max(0, n - dst)]
        if rng.random() < 0.1 and n > 1000:  # a run
            s0 = rng.randrange(0, n - 1000)
            b[s0:s0 + 900] = bytes([rng.randrange(256)]) * 900
        blocks.append(bytes(b[:n]))
    check_blocks(gpu_plugin, oracle, blocks, level)


@pytest.mark.parametrize("level", [1, 2, 3, 4, 6, 9, 12, 0x101, 0x105])
def test_segment_work_items(gpu_plugin, oracle, level):
    """qzstd_hip_block_t.parseFrom (every level): a work item that holds a block up to a segment's end and parses the
    segment(s) only (what the per-block paths submit: up to 32 items per 128 KiB block) — bit-exact against the oracle's
    qzo_find_sequences_from, ragged sizes included; a parseFrom that is no segment boundary is refused"""
    items, froms = [], []
    for gen, size in (("text", 131072), ("system", 131072), ("weblog", 100001), ("binary", 70000), ("mix", 33000)):
        blk = K.by_name(gen, size, seed=29)
        for s0 in range(0, len(blk), 32768):
            items.append(blk[:min(len(blk), s0 + 32768)])
            froms.append(s0)
    # items that end right behind a segment boundary (the fast-forward over the history reads 16 bytes past parseFrom - 16;
    # the tile loop starts with a ring prefilled from the middle of the block), degenerate content included
    base = K.by_name("system", 131072, seed=31)
    for s0 in (32768, 65536, 98304):
        for extra in (1, 2, 7, 15, 16, 17, 31, 511, 512, 513, 4607, 4608, 4609):
            items.append(base[:s0 + extra])
            froms.append(s0)
    for blk in (bytes(131072), b"ab" * 65536, (b"abcdefgh" * 5 + b"X") * 3197):
        for s0 in (32768, 98304):
            items.append(blk[:min(len(blk), s0 + 32768)])
            froms.append(s0)
    # the service path's granularity: one item per 4 KiB segment (32 per 128 KiB block), and runs of three segments
    fine = K.by_name("system", 131072, seed=37)
    for step in (4096, 12288):
        for s0 in range(0, len(fine), step):
            items.append(fine[:min(len(fine), s0 + step)])
            froms.append(s0)
    ragged = K.by_name("weblog", 50001, seed=41)
    for s0 in range(0, len(ragged), 4096):
        items.append(ragged[:min(len(ragged), s0 + 4096)])
        froms.append(s0)
    counts, seqs, stride = gpu_plugin.find_batch(items, level, parse_from=froms)
    for i, (blk, s0) in enumerate(zip(items, froms)):
        want_n, want = oracle.find(oracle.profile(level, len(blk)), blk, cap=stride, parse_from=s0)
        assert counts[i] == want_n, "item %d (len %d from %d): %d sequences, oracle %d" % (i, len(blk), s0, counts[i], want_n)
        assert np.array_equal(seqs_to_np(seqs, i * stride, want_n), seqs_to_np(want, 0, want_n)), "item %d differs" % i
    blk = K.text(3, 70000)
    counts, _, _ = gpu_plugin.find_batch([blk, blk, blk], level, parse_from=[1000, 98304, 32768])
    assert counts[0] == B.NSEQ_ERROR and counts[1] == B.NSEQ_ERROR and counts[2] != B.NSEQ_ERROR


def test_chain_insert_ballot_path_in_a_child_process():
    """The chain levels insert a tile with one returning ds_max per window where the device's LDS serves same-address lanes in
    lane order (probed once per device).  The ballot path it replaces stays as the fallback: force it in a fresh process
    (QZSTD_HIP_ORDERED_LDS=0 is read at the first launch) and compare with the oracle there too — degenerate blocks included,
    where every lane of a window hits one slot."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np
import qz_bind as B, qz_corpus as K
plug, orc = B.Plugin(B.PLUGIN_SO), B.Oracle()
data = K.by_name("system", 6 * 131072, seed=3)
blocks = [data[o:o + 131072] for o in range(0, len(data), 131072)] + [bytes(131072), b"ab" * 50000, (b"abcdefgh" * 5 + b"X") * 3000,
          K.by_name("weblog", 32768, seed=5)]
for level in (5, 6, 9, 12):
    counts, seqs, stride = plug.find_batch(blocks, level)
    a = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)
    for i, blk in enumerate(blocks):
        n, want = orc.find(orc.profile(level, len(blk)), blk, cap=stride)
        assert counts[i] == n, (level, i, counts[i], n)
        w = np.frombuffer(want, dtype=np.uint32).reshape(-1, 4)[:n, :3]
        assert np.array_equal(a[i * stride:i * stride + n, :3], w), (level, i)
print("ballot path ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, QZSTD_HIP_ORDERED_LDS="0"))
    assert out.returncode == 0 and "ballot path ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("level", [1, 3, 6, 12, 0x106])
def test_near_kernels_at_the_ring_boundary(gpu_plugin, oracle, level):
    """round 4: a launch whose blocks all fit the LDS ring (maxBlockLen <= 32 KiB) runs the NEAR kernels — every source is compared from LDS, the
    device-memory side of the compares is compiled out; one byte more and the launch takes the kernels with both sides.  Same sequences either
    way: the same blocks alone (NEAR), next to a 32 769-byte block (not NEAR), and with sources as far back as a 32 KiB block allows"""
    far = K.by_name("text", 3000, seed=5)
    blocks = [K.by_name("weblog", 32768, seed=4), K.by_name("system", 32768, seed=9), K.by_name("mix", 32767, seed=2),
              far + K.incompressible(3, 32768 - 6000) + far,          # the only matches lie ~29.7 KiB back: beyond kNear, inside the ring
              K.by_name("text", 20000, seed=1)]
    check_blocks(gpu_plugin, oracle, blocks, level)                                       # maxBlockLen 32768: NEAR
    check_blocks(gpu_plugin, oracle, blocks + [K.by_name("text", 32769, seed=8)], level)  # maxBlockLen 32769: not NEAR


@pytest.mark.parametrize("level", [1, 3, 0x101])
def test_descriptor_longer_than_the_launch_says_is_refused(gpu_plugin, oracle, level):
    """round 6: below the chain levels a launch parses out of a scratch region sized by its maxBlockLen (one parse word per position); a descriptor
    longer than that must come back as an error block — never parsed out of words that lie in its neighbour's region — and the blocks that fit
    must be the oracle's"""
    blocks = [K.by_name("text", 70000, seed=3), K.by_name("system", 40000, seed=4), K.by_name("mix", 39999, seed=5)]
    counts, seqs, stride = gpu_plugin.find_batch(blocks, level, launch_max_len=40000)
    assert counts[0] == B.NSEQ_ERROR
    for i in (1, 2):
        want_n, want = oracle.find(oracle.profile(level, len(blocks[i])), blocks[i], cap=stride)
        assert counts[i] == want_n
        assert np.array_equal(seqs_to_np(seqs, i * stride, want_n), seqs_to_np(want, 0, want_n))


def test_two_workgroups_per_cu_where_the_design_says_so(gpu_plugin):
    """the LDS budget is sized for two workgroups per CU at levels 1-2 and 5-12 and one at levels 3-4 (qzstd_hip_lds_bytes); the runtime's
    occupancy query has to agree — a register or LDS regression that halves the residency shows here, not only in the timings"""
    L = gpu_plugin.lib
    for level, want in ((1, 2), (2, 2), (3, 1), (4, 1), (5, 2), (6, 2), (9, 2), (12, 2), (0x101, 2), (0x106, 2)):
        assert L.qzstd_hip_occupancy(0, level) == want, (level, L.qzstd_hip_occupancy(0, level), gpu_plugin.err())


def test_copy_in_kernel_moves_pinned_bytes_to_device_memory(gpu_plugin):
    """qzstd_hip_copy_in (round 4: an announcement's staging copy reaches device memory by a kernel on the launch's stream, not by the runtime's
    copy path): byte-exact for sizes from one 16-byte unit to several MiB — fewer units than lanes, a grid-stride tail, more units than the
    1024-workgroup grid covers in one step — ordered with what follows on the stream; unaligned or odd-sized requests are refused"""
    L = gpu_plugin.lib
    L.qzstd_hip_copy_in.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.qzstd_hip_host_device_ptr.argtypes = [C.c_void_p]
    L.qzstd_hip_host_device_ptr.restype = C.c_void_p
    cap = 20 << 20
    h_in, h_out = L.qzstd_hip_host_alloc(cap), L.qzstd_hip_host_alloc(cap)
    d = L.qzstd_hip_malloc(0, cap)
    assert h_in and h_out and d, gpu_plugin.err()
    try:
        dv = L.qzstd_hip_host_device_ptr(h_in)
        assert dv, gpu_plugin.err()
        data = K.by_name("mix", cap, seed=77)
        C.memmove(h_in, data, cap)
        for n in (16, 48, 4096, 16384 + 16, 1 << 20, (1 << 24) + 4096 + 16, cap):
            C.memset(h_out, 0xEE, n)
            assert L.qzstd_hip_copy_in(0, None, d, dv, n) == 0, gpu_plugin.err()
            assert L.qzstd_hip_memcpy_d2h(0, None, h_out, d, n) == 0, gpu_plugin.err()  # (same stream: ordered behind the copy kernel)
            assert L.qzstd_hip_stream_sync(0, None) == 0, gpu_plugin.err()
            assert C.string_at(h_out, n) == data[:n], "copy of %d bytes differs" % n
        # from an offset inside the pinned buffer (a part of an announcement starts at its first block)
        assert L.qzstd_hip_copy_in(0, None, d, dv + 131072 * 3, 131072) == 0 and L.qzstd_hip_memcpy_d2h(0, None, h_out, d, 131072) == 0
        assert L.qzstd_hip_stream_sync(0, None) == 0 and C.string_at(h_out, 131072) == data[131072 * 3:131072 * 4]
        assert L.qzstd_hip_copy_in(0, None, d, dv, 0) == 0
        assert L.qzstd_hip_copy_in(0, None, d, dv, 24) != 0 and L.qzstd_hip_copy_in(0, None, d, dv + 8, 32) != 0 and L.qzstd_hip_copy_in(0, None, None, dv, 32) != 0
    finally:
        L.qzstd_hip_free(0, d)
        L.qzstd_hip_host_free(h_in)
        L.qzstd_hip_host_free(h_out)
