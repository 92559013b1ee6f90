#!/usr/bin/env python3
"""Design-time experiment: compressed size of the oracle's sequences (through libzstd's own
entropy stage) vs libzstd's software match-finder, benchmark.c framing (one frame per chunk).

usage: tools/ratio_probe.py [--level 1] [--chunk 131072] [--mb 8] [--set k=v ...]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B  # noqa: E402
import qz_corpus as K  # noqa: E402


def run(z, orc, data, chunk, level, prof, ext_rep=None, verify=True):
    zc = z.cctx(level, producer=orc.producer_addr, state=B.C.addressof(prof) if prof is not None else None,
                ext_repcodes=ext_rep)
    t = time.time()
    total, frames = z.compress_chunks(zc, data, chunk)
    dt = time.time() - t
    z.free(zc)
    if verify:
        o = 0
        for f in frames[:: max(1, len(frames) // 16)]:
            pass
        out = b"".join(z.decompress(f, chunk) for f in frames)
        assert out == data, "round trip mismatch"
    return total, dt


def sw(z, data, chunk, level):
    zc = z.cctx(level)
    total, _ = z.compress_chunks(zc, data, chunk)
    z.free(zc)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=131072)
    ap.add_argument("--mb", type=float, default=4)
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--sweep", default="")
    ap.add_argument("--ext-rep", type=int, default=None)
    a = ap.parse_args()
    z = B.Zstd()
    orc = B.Oracle(os.environ.get("QZ_ORACLE_SO", B.ORACLE_SO))
    per = int(a.mb * K.MiB)
    corp = {}
    for label, part in K.system_corpus_parts():
        corp[label] = part[:per]
    corp["syn_text"] = K.text(1, per)
    corp["syn_binary"] = K.binary_struct(2, per)
    corp["syn_weblog"] = K.weblog(4, per)
    corp["syn_mixent"] = K.mixed_entropy(5, per)

    base = orc.profile(a.level, a.chunk)
    for kv in a.set:
        k, v = kv.split("=")
        setattr(base, k, int(v))
    variants = [("base", {})]
    if a.sweep:
        k, vals = a.sweep.split("=")
        variants = [("%s=%s" % (k, v), {k: int(v)}) for v in vals.split(",")]
    print("profile:", base.as_dict())
    sws = {l: sw(z, d, a.chunk, a.level) for l, d in corp.items()}
    hdr = "%-12s %9s %9s" % ("corpus", "bytes", "sw")
    for name, _ in variants:
        hdr += " %14s" % name
    print(hdr)
    tot_sw = 0
    tot = [0] * len(variants)
    for l, d in corp.items():
        row = "%-12s %9d %9d" % (l, len(d), sws[l])
        tot_sw += sws[l]
        for i, (name, kv) in enumerate(variants):
            p = B.OracleProfile.from_buffer_copy(base)
            for k, v in kv.items():
                setattr(p, k, v)
            c, dt = run(z, orc, d, a.chunk, a.level, p, a.ext_rep)
            tot[i] += c
            row += " %8d %5.3f" % (c, sws[l] / c)
        print(row)
    row = "%-12s %9s %9d" % ("TOTAL", "", tot_sw)
    for i in range(len(variants)):
        row += " %8d %5.3f" % (tot[i], tot_sw / tot[i])
    print(row)
    print("(second number = sw_size / oracle_size; >= 0.98 means within 2 %)")


if __name__ == "__main__":
    main()
