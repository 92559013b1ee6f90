#!/usr/bin/env python3
"""Pins the oracle's LZ4s decoder restatement (qzo_lz4s_decode) to the REFERENCE'S OWN CODE (round-3 verdict, item 8).

The only self-contained arithmetic of the reference's path is QZSTD_decLz4s (/root/reference/src/qatseqprod.c:1013-1091, with the
macros of :99-104 and the little-endian readers of :966-990); everything else needs the QAT SDK and QAT silicon.  This script — run
in the BUILD CONTAINER only, where /root/reference exists —

  1. lifts exactly those line ranges out of /root/reference/src/qatseqprod.c into a scratch translation unit in a temporary
     directory (nothing of the reference's text is written into the repo, and the scratch directory is deleted afterwards);
  2. compiles it with gcc against the image's own zstd.h (1.4.9, /opt/conda/include: it has ZSTD_Sequence under
     ZSTD_STATIC_LINKING_ONLY).  Two names the lifted lines use come from outside them and are given on the command line:
     -DZSTD_SEQUENCE_PRODUCER_ERROR='((size_t)(-1))' (zstd >= 1.5.4's public constant; the image has no such header) and a
     QZSTD_LOG that expands to nothing (the reference's logging macro, :202-206);
  3. runs the reference's decoder on seeded random LZ4s streams — well-formed ones of every shape the format has (literal-run and
     match-length extensions, stored match length 0 = "no match, merge the literals", matches longer than 65535 that the
     reference truncates to 16 bits, empty and literal-only streams), capacity-rule cases, and streams whose last literal run
     overshoots the buffer (decode error) — and
  4. writes inputs and the reference's outputs to tests/golden/lz4s_ref_vectors.json.  A fixture is data: inputs and expected
     outputs, no reference text.

  5. (round 5) ORACLE vectors: the oracle's own sequences of seeded blocks (every offset < 65536: LZ4s' range), serialised as LZ4s
     by serialise_lz4s() below, decoded by the same lifted reference decoder: the reference's code certifies the oracle's emission
     rules on REAL output of this match-finder (minimum match, delimiter, literal carry-over), not only on synthetic streams.  One
     vector runs the oracle without segment boundaries (profile.segLog = 0, extLog = 17) on a 128 KiB periodic block: a match longer than 65535,
     which the reference truncates to 16 bits (:1062).

tests/test_oracle_golden.py::test_lz4s_decoder_matches_the_reference_decoder then requires qzo_lz4s_decode to agree on every
vector (CPU suite; it reads only the committed JSON — /root/reference does not exist on the GPU box).

usage: python tools/make_lz4s_golden.py [--check]     (--check: regenerate in memory and compare with the committed file)
"""
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/qatseqprod.c"
OUT = os.path.join(ROOT, "tests", "golden", "lz4s_ref_vectors.json")
ZSTD_INC = "/opt/conda/include"
# 1-based inclusive line ranges of the reference file: the LZ4s macros, the LE16 readers, the decoder
RANGES = ((99, 104), (966, 990), (1013, 1091))
ANCHORS = {99: "#define ML_BITS 4", 104: "#define LZ4MINMATCH 2", 966: "static unsigned isLittleEndian(void)",
           1013: "static size_t QZSTD_decLz4s(ZSTD_Sequence *outSeqs, size_t outSeqsCapacity,", 1091: "}"}

HARNESS = r'''
#include <stdio.h>
#include <stdlib.h>
/* stdin: one vector per line "cap hex"; stdout: "E" or "n off lit ml off lit ml ..." */
int main(void)
{
    static char line[1 << 20];
    while (fgets(line, sizeof(line), stdin)) {
        size_t cap, n = 0, i;
        char *p = line;
        unsigned char *buf;
        ZSTD_Sequence *out;
        size_t r;
        cap = strtoul(p, &p, 10);
        while (*p == ' ') p++;
        buf = (unsigned char *)calloc(1, (1 << 19) + 4096); /* padded: streams that overshoot are read, never out of bounds */
        while (p[0] && p[1] && p[0] != '\n') { unsigned v; sscanf(p, "%2x", &v); buf[n++] = (unsigned char)v; p += 2; }
        out = (ZSTD_Sequence *)calloc(cap + 8, sizeof(ZSTD_Sequence));
        r = QZSTD_decLz4s(out, cap, buf, (unsigned int)n);
        if (r == ZSTD_SEQUENCE_PRODUCER_ERROR) printf("E\n");
        else {
            printf("%zu", r);
            for (i = 0; i < r; i++) printf(" %u %u %u", out[i].offset, out[i].litLength, out[i].matchLength);
            printf("\n");
        }
        free(buf); free(out);
    }
    return 0;
}
'''


def build_reference_decoder(tmp):
    lines = open(REF, encoding="utf-8", errors="replace").read().split("\n")
    for ln, text in ANCHORS.items():
        if lines[ln - 1].strip() != text.strip():
            raise SystemExit("reference line %d is not %r: the line ranges of this script no longer fit %s" % (ln, text, REF))
    tu = os.path.join(tmp, "declz4s_scratch.c")
    with open(tu, "w") as f:
        f.write("#define ZSTD_STATIC_LINKING_ONLY\n#include <zstd.h>\n#include <string.h>\n#include <stddef.h>\n")
        for a, b in RANGES:
            f.write("\n".join(lines[a - 1:b]) + "\n")
        f.write(HARNESS)
    exe = os.path.join(tmp, "declz4s")
    subprocess.check_call(["gcc", "-O1", "-w", "-I" + ZSTD_INC, "-DZSTD_SEQUENCE_PRODUCER_ERROR=((size_t)(-1))", "-DQZSTD_LOG(...)=",
                           "-o", exe, tu])
    return exe


def length_bytes(rest):
    """the 255-run extension of a nibble that was 15: bytes adding up to `rest`"""
    out = bytearray()
    while rest >= 255:
        out.append(255)
        rest -= 255
    out.append(rest)
    return bytes(out)


def token_seq(rng, lit, stored_ml, offset):
    """one LZ4s sequence: token, literal-run extension, literals, LE16 offset, match-length extension"""
    b = bytearray()
    b.append((min(lit, 15) << 4) | min(stored_ml, 15))
    if lit >= 15:
        b += length_bytes(lit - 15)
    b += bytes(rng.randrange(256) for _ in range(lit))
    b += bytes((offset & 0xFF, offset >> 8))
    if stored_ml >= 15:
        b += length_bytes(stored_ml - 15)
    return bytes(b)


def last_token(rng, lit):
    b = bytearray([min(lit, 15) << 4])
    if lit >= 15:
        b += length_bytes(lit - 15)
    b += bytes(rng.randrange(256) for _ in range(lit))
    return bytes(b)


def make_streams(seed=20260929):
    rng = random.Random(seed)
    vecs = []

    def add(name, raw, cap=64):
        vecs.append({"name": name, "cap": cap, "hex": raw.hex()})

    add("empty", b"")
    add("one_empty_token", b"\x00")
    add("literals_only_7", last_token(rng, 7))
    add("literals_only_300", last_token(rng, 300), cap=8)
    for i in range(24):  # well-formed random streams
        nseq = rng.randrange(1, 9)
        raw = b""
        for _ in range(nseq):
            lit = rng.choice((0, 0, 1, 3, 14, 15, 16, 40, 254 + 15, 255 + 15, 300))
            ml = rng.choice((0, 1, 2, 5, 14, 15, 16, 100, 255 + 15, 254 + 15, 700))  # stored; 0 = no match (literals carried over)
            raw += token_seq(rng, lit, ml, rng.randrange(1, 65536))
        raw += last_token(rng, rng.choice((0, 0, 2, 15, 33)))
        add("random_%02d" % i, raw)
    # matches the reference truncates to 16 bits: stored + 2 >= 65536
    add("match_65534_plus_2_wraps_to_0", token_seq(rng, 2, 65534, 9) + last_token(rng, 1))
    add("match_65600_truncated", token_seq(rng, 0, 65600, 1234) + last_token(rng, 0))
    add("match_65533_is_65535", token_seq(rng, 1, 65533, 77) + last_token(rng, 3))
    # runs of "no match" tokens: their literals pile up in the next sequence, or in the delimiter
    add("three_no_match_then_match", token_seq(rng, 4, 0, 1) + token_seq(rng, 0, 0, 2) + token_seq(rng, 20, 0, 3) + token_seq(rng, 1, 6, 300) + last_token(rng, 2))
    add("no_match_then_end", token_seq(rng, 9, 0, 5) + last_token(rng, 4))
    # capacity rule (:1073-1076): seqsIdx >= cap - 1 is an error
    five = b"".join(token_seq(rng, 1, 3, 10 + k) for k in range(5)) + last_token(rng, 0)
    for cap in (4, 5, 6, 7, 8):
        add("five_matches_cap_%d" % cap, five, cap=cap)
    # decode errors: the last literal run overshoots the buffer (ip != endip, :1086-1089); a stream that ends inside a match field
    add("literal_run_overshoots", token_seq(rng, 1, 3, 9) + bytes([0x50, 1, 2]))
    add("ends_before_offset", token_seq(rng, 2, 4, 9) + bytes([0x21, 7, 7, 0x10]))
    return vecs


def serialise_lz4s(seqs, block: bytes) -> bytes:
    """(offset, litLength, matchLength) triples incl. the delimiter + the block's literals -> an LZ4s stream (token, literal-run
    extension, literals, LE16 offset, match-length extension; stored match length = matchLength - 2: LZ4MINMATCH, reference :104,:1061)"""
    out = bytearray()
    pos = 0
    for off, lit, ml in seqs[:-1]:
        stored = ml - 2
        assert 0 < off < 65536 and stored >= 1
        out.append((min(lit, 15) << 4) | min(stored, 15))
        if lit >= 15:
            out += length_bytes(lit - 15)
        out += block[pos:pos + lit]
        out += bytes((off & 0xFF, off >> 8))
        if stored >= 15:
            out += length_bytes(stored - 15)
        pos += lit + ml
    off, lit, ml = seqs[-1]
    assert off == 0 and ml == 0 and pos + lit == len(block)
    out.append(min(lit, 15) << 4)
    if lit >= 15:
        out += length_bytes(lit - 15)
    out += block[pos:pos + lit]
    return bytes(out)


# (name, generator of tools/qz_corpus.py, seed, bytes, level, segLog override or None)
ORACLE_CASES = (("oracle_text_8k_L1", "text", 21, 8192, 1, None), ("oracle_weblog_8k_L1", "weblog", 22, 8192, 1, None),
                ("oracle_binary_8k_L3", "binary", 23, 8192, 3, None), ("oracle_text_6000_L6", "text", 24, 6000, 6, None),
                ("oracle_weblog_8k_L12", "weblog", 25, 8192, 12, None), ("oracle_mix_16k_L1", "mix", 26, 16384, 1, None),
                ("oracle_mixed_entropy_8k_L3", "mixed_entropy", 27, 8192, 3, None), ("oracle_random_2k_L1", "random", 28, 2048, 1, None),
                ("oracle_periodic_128k_L1_no_segments", "periodic", 29, 131072, 1, 0))


def oracle_block(gen: str, seed: int, size: int) -> bytes:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import qz_corpus as K
    if gen == "periodic":  # 37 seeded bytes over and over: one long match at offset 37
        rng = random.Random(seed)
        unit = bytes(rng.randrange(256) for _ in range(37))
        return (unit * (size // 37 + 1))[:size]
    return K.by_name(gen, size, seed)


def oracle_sequences(gen: str, seed: int, size: int, level: int, seglog):
    """the oracle's sequences of the case's block: [(offset, litLength, matchLength), ...] incl. the delimiter"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import qz_bind as B
    orc = B.Oracle()
    blk = oracle_block(gen, seed, size)
    prof = orc.profile(level, len(blk))
    if seglog is not None:  # no segment boundaries, and the extension's own bound (profile.extLog) out of the way: the match may run to the block's end
        prof.segLog = seglog
        prof.extLog = 17
    n, seqs = orc.find(prof, blk)
    assert n != B.SEQ_ERROR and n >= 1
    return blk, [(seqs[i].offset, seqs[i].litLength, seqs[i].matchLength) for i in range(n)]


def make_oracle_streams():
    vecs = []
    for name, gen, seed, size, level, seglog in ORACLE_CASES:
        blk, seqs = oracle_sequences(gen, seed, size, level, seglog)
        vecs.append({"name": name, "cap": len(seqs) + 8, "gen": gen, "seed": seed, "size": size, "level": level, "seglog": seglog,
                     "oracle_sequences": len(seqs), "hex": serialise_lz4s(seqs, blk).hex()})
    return vecs


def run_reference(vecs):
    tmp = tempfile.mkdtemp(prefix="qz_lz4s_")
    try:
        exe = build_reference_decoder(tmp)
        inp = "".join("%d %s\n" % (v["cap"], v["hex"]) for v in vecs)
        out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split("\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)  # the scratch TU (reference text) never outlives the run
    for v, line in zip(vecs, out):
        if line.strip() == "E":
            v["expect"] = "error"
        else:
            w = [int(x) for x in line.split()]
            v["expect"] = [w[1 + 3 * i:4 + 3 * i] for i in range(w[0])]
    return vecs


def main():
    if not os.path.isfile(REF):
        raise SystemExit("%s is absent: this generator runs in the build container only (the committed JSON is what travels)" % REF)
    doc = {"source": "outputs of the reference's own QZSTD_decLz4s (src/qatseqprod.c:1013-1091, compiled from the reference tree by "
                     "tools/make_lz4s_golden.py with the image's zstd.h 1.4.9) on seeded random LZ4s streams; expect = [offset, litLength, "
                     "matchLength] per sequence incl. the delimiter, or \"error\" (ZSTD_SEQUENCE_PRODUCER_ERROR)",
           "generator": "tools/make_lz4s_golden.py", "seed": 20260929, "vectors": run_reference(make_streams()),
           "oracle_source": "the ORACLE's sequences (oracle/qzstd_oracle.c, qzo_find_sequences) of seeded blocks — tools/qz_corpus.py generators, "
                            "the case's seed / size / level — serialised as LZ4s by serialise_lz4s() and decoded by the same reference decoder: "
                            "expect is what the reference's code makes of this match-finder's real output",
           "oracle_vectors": run_reference(make_oracle_streams())}
    # one vector per line (the expected lists of the oracle vectors hold hundreds of triples each: indent=1 made a 17 000-line file of them)
    head = {k: v for k, v in doc.items() if k not in ("vectors", "oracle_vectors")}
    text = "{\n" + "".join(" %s: %s,\n" % (json.dumps(k), json.dumps(v)) for k, v in head.items())
    text += ' "vectors": [\n' + ",\n".join("  " + json.dumps(v) for v in doc["vectors"]) + "\n ],\n"
    text += ' "oracle_vectors": [\n' + ",\n".join("  " + json.dumps(v) for v in doc["oracle_vectors"]) + "\n ]\n}\n"
    assert json.loads(text) == doc
    if "--check" in sys.argv:
        same = open(OUT).read() == text
        print("lz4s_ref_vectors.json %s what the reference's decoder produces now" % ("equals" if same else "DIFFERS FROM"))
        sys.exit(0 if same else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s: %d vectors, %d of them decode errors; %d oracle vectors" % (OUT, len(doc["vectors"]), sum(v["expect"] == "error" for v in doc["vectors"]),
                                                                               len(doc["oracle_vectors"])))


if __name__ == "__main__":
    main()
