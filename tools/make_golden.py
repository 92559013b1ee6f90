#!/usr/bin/env python3
"""Generates tests/golden/*.json (run in the build container; commit the output).

Two fixture families, both DATA (inputs are seeds, outputs are numbers / hex strings):

* sw_sizes.json   — per (generator, seed, blockSize, level): software libzstd 1.5.7 compressed
                    size (one frame per block, reference test/benchmark.c:300-321 framing), the
                    size libzstd produces from the ORACLE's sequences (registered as the sequence
                    producer, ZSTD_c_validateSequences=1), the oracle's nbSeq / sum(matchLength) /
                    FNV-1a of all (offset, litLength, matchLength), and a CRC32 of the input.
                    Pins the oracle against libzstd (round trip + size) and against itself.
* lz4s_vectors.json — hand-assembled LZ4s streams and the ZSTD_Sequence arrays the reference's
                    decoder semantics (QZSTD_decLz4s, src/qatseqprod.c:1013-1091) give for them.
                    Written out by hand from the format rules, NOT produced by running code.
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import qz_bind as B  # noqa: E402
import qz_corpus as K  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = [  # (generator, seed, blockSize, nBlocks, level)
    ("text", 1, 131072, 4, 1), ("text", 3, 131072, 2, 6), ("binary", 2, 131072, 4, 1),
    ("weblog", 4, 32768, 8, 12), ("weblog", 4, 131072, 2, 1), ("mixed_entropy", 5, 131072, 5, 3),
    ("mix", 2, 131072, 6, 1), ("mix", 2, 65536, 6, 1), ("random", 9, 131072, 1, 1), ("text", 7, 100001, 1, 1),
    # level | 0x100 = the caller compresses with ZSTD_c_searchForExternalRepcodes on: repeat-offset aware parse
    ("text", 1, 131072, 2, 0x101), ("binary", 2, 131072, 3, 0x106), ("mix", 2, 131072, 4, 0x103), ("weblog", 4, 32768, 4, 0x102),
]
REP = 0x100


def oracle_cctx(z, orc, level, block):
    """CCtx with the oracle registered; a flagged level passes its profile explicitly and turns the
    external repcode search on (what a caller of the plugin does together with QZSTD_HIP_EXT_REPCODES=1)"""
    if level & REP:
        prof = orc.profile(level, block)
        return z.cctx(level & ~REP, producer=orc.producer_addr, state=B.C.addressof(prof), validate=True, ext_repcodes=1), prof
    return z.cctx(level, producer=orc.producer_addr, state=None, validate=True), None


def main():
    z, orc = B.Zstd(), B.Oracle()
    rows = []
    for gen, seed, block, nb, level in CASES:
        data = K.by_name(gen, block * nb, seed)
        zc = z.cctx(level & ~REP)
        sw, _ = z.compress_chunks(zc, data, block)
        z.free(zc)
        zc, keep = oracle_cctx(z, orc, level, block)
        got, frames = z.compress_chunks(zc, data, block)
        z.free(zc)
        assert b"".join(z.decompress(f, block) for f in frames) == data
        nseq = summ = 0
        fnv = []
        for o in range(0, len(data), block):
            blk = data[o:o + block]
            n, seqs = orc.find(orc.profile(level, len(blk)), blk)
            m, l, h = orc.stats(seqs, n)
            assert m + l == len(blk)
            nseq += n
            summ += m
            fnv.append("%016x" % h)
        rows.append({"gen": gen, "seed": seed, "block": block, "blocks": nb, "level": level,
                     "crc32": zlib.crc32(data) & 0xFFFFFFFF, "sw_size": sw, "oracle_size": got,
                     "oracle_nseq": nseq, "oracle_sum_match": summ, "oracle_fnv": fnv})
        print(rows[-1]["gen"], block, level, "sw", sw, "oracle", got, "%.4f" % (sw / got))
    with open(os.path.join(OUT, "sw_sizes.json"), "w") as f:
        json.dump({"zstd": z.version(), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
