#!/usr/bin/env python3
"""Debug helper (GPU box): first differing sequence between the HIP path and the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import qz_bind as B, qz_corpus as K

def main():
    plug, orc = B.Plugin(os.environ.get("QZ_PLUGIN_SO", B.PLUGIN_SO)), B.Oracle()
    level = int(os.environ.get("QZ_LEVEL", "1"), 0)  # e.g. 0x106 = level 6 | QZSTD_HIP_LEVEL_REPCODES
    sizes = [int(a) for a in sys.argv[1:]] or [600, 1500, 3000, 9000, 131072]
    corpus = os.environ.get("QZ_CORPUS", "text")
    off0 = int(os.environ.get("QZ_OFFSET", "0"))
    degen = [bytes(131072), b"\xff" * 70000, b"ab" * 50000, b"abc" * 40000, (b"0123456789" * 13108)[:131072],
             bytes(range(256)) * 512, b"x" + bytes(5000), K.incompressible(9, 131072),
             (b"abcdefgh" * 5 + b"X") * 3000, b"".join(b"record%05d;" % (i % 97) + bytes(53) for i in range(2000))]
    if corpus == "records":
        base = degen[9]
    elif corpus.startswith("degen"):
        base = degen[int(corpus[5:])]
    else:
        base = K.by_name(corpus, off0 + 140000)[off0:]
    for n in sizes:
        blk = base[:n]
        counts, seqs, stride = plug.find_batch([blk], level)
        want_n, want = orc.find(orc.profile(level, n), blk, cap=stride)
        g = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)[:max(counts[0], 1) if counts[0] != B.NSEQ_ERROR else 1, :3]
        w = np.frombuffer(want, dtype=np.uint32).reshape(-1, 4)[:want_n, :3]
        m = min(len(g), len(w))
        diff = np.nonzero((g[:m] != w[:m]).any(axis=1))[0]
        print("n=%d gpu_count=%d oracle_count=%d first_diff=%s" % (n, counts[0], want_n, diff[:1]), flush=True)
        if len(diff):
            i = int(diff[0])
            pos_g = int(g[:i, 1].sum() + g[:i, 2].sum())
            print("  at seq %d (block pos %d, window %d lane %d):" % (i, pos_g, pos_g // 64, pos_g % 64))
            for k in range(max(0, i - 2), min(m, i + 4)):
                print("   %5d gpu %-22s oracle %-22s" % (k, g[k].tolist(), w[k].tolist()))
def timing():
    plug = B.Plugin(os.environ.get("QZ_PLUGIN_SO", os.path.join(B.PKG_DIR, "lib", "libqatseqprod_dbg.so")))
    data = K.system_corpus(int(os.environ.get("QZ_BLOCKS", "256")) * 131072)[0]  # 256 = one workgroup per CU, 512 = two
    blocks = [data[o:o + 131072] for o in range(0, len(data), 131072)]
    pf = int(os.environ.get("QZ_PARSE_FROM", "0"))  # segment items: the tiles before this position are history
    counts, seqs, stride = plug.find_batch(blocks, int(os.environ.get("QZ_LEVEL", "1"), 0), parse_from=[pf] * len(blocks) if pf else None)
    a = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)
    rows = np.array([a[(i + 1) * stride - 1] for i in range(len(blocks))], dtype=np.float64)
    c = np.array(counts, dtype=np.float64)
    print("blocks %d  seq/block %.0f" % (len(blocks), c.mean()))
    rows2 = np.array([a[(i + 1) * stride - 2] for i in range(len(blocks))], dtype=np.float64)
    print("parse wave per tile: I1 %.0f waitB1 %.0f I2 %.0f waitB2 %.0f" % tuple(rows2.mean(axis=0) / 256))
    if os.environ.get("QZ_HWID"):
        # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13]
        hw = np.array([[a[(i + 1) * stride - 12 - wv][0] for wv in range(9)] for i in range(len(blocks))], dtype=np.int64)
        simd = (hw >> 4) & 3
        import collections
        print("SIMD of waves 0..8 (matchers 0-7, parse wave 8), most common patterns:")
        for pat, cnt in collections.Counter(tuple(r) for r in simd.tolist()).most_common(6):
            print("   ", pat, cnt)
        cu = ((hw[:, 0] >> 8) & 15) | (((hw[:, 0] >> 13) & 7) << 4) | (((hw[:, 0] >> 12) & 1) << 7)
        print("parse-wave SIMD histogram:", np.bincount(simd[:, 8], minlength=4).tolist())
    if int(os.environ.get("QZ_LEVEL", "1"), 0) & 0xFF >= 5:
        ntile = 256.0
        for wv in range(8):
            c0 = np.array([a[(i + 1) * stride - 24 - 2 * wv] for i in range(len(blocks))], dtype=np.float64) * 16
            c1 = np.array([a[(i + 1) * stride - 25 - 2 * wv] for i in range(len(blocks))], dtype=np.float64)
            m0, m1 = c0.mean(axis=0) / ntile, c1.mean(axis=0)
            print("matcher wave %d chain walk per tile: entry build %.0f | per tile over %.1f steps: fetch issue %.0f, tests %.0f, heads+extensions %.0f, wait for next entry %.0f"
                  % (wv, m0[0], m1[1] / ntile, m0[1], m0[2], m0[3], m1[0] * 16 / ntile))
    if os.environ.get("QZ_POST"):  # the deferred parse (levels 1-4 of a launch): cycles per BLOCK and wave
        for wv in range(8):
            rw = np.array([a[(i + 1) * stride - 44 - wv] for i in range(len(blocks))], dtype=np.float64)
            print("matcher wave %d after the loop, per block: wait for the loop's last wave %.0f, pass 1 (parse) %.0f, wait for its last wave %.0f, pass 2 (emission) %.0f"
                  % ((wv,) + tuple(rw.mean(axis=0))))
    for wv in range(8):
        rw = np.array([a[(i + 1) * stride - 3 - wv] for i in range(len(blocks))], dtype=np.float64)
        print("matcher wave %d per tile: I1 %.0f waitB1 %.0f I2 %.0f waitB2 %.0f" % ((wv,) + tuple(rw.mean(axis=0) / 256)))


if os.environ.get("QZ_TIMING"):
    timing()
else:
    main()
