#!/usr/bin/env python3
"""ratio of oracle-produced frames vs software zstd on the exact sample bench.py uses
(first N MiB of the system corpus, 128 KiB frames)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B, qz_corpus as K
ap = argparse.ArgumentParser(); ap.add_argument("--mb", type=int, default=32); ap.add_argument("--level", type=int, default=1)
ap.add_argument("--set", nargs="*", default=[]); ap.add_argument("--sweep", default=""); ap.add_argument("--ext-rep", type=int, default=None)
a = ap.parse_args()
z, orc = B.Zstd(), B.Oracle()
data = K.system_corpus(a.mb * K.MiB)[0]
zc = z.cctx(a.level); sw, _ = z.compress_chunks(zc, data, 131072); z.free(zc)
base = orc.profile(a.level, 131072)
for kv in a.set:
    k, v = kv.split("="); setattr(base, k, int(v))
variants = [("base", {})]
if a.sweep:
    k, vals = a.sweep.split("="); variants = [("%s=%s" % (k, v), {k: int(v)}) for v in vals.split(",")]
print("sw", sw)
for name, kv in variants:
    p = B.OracleProfile.from_buffer_copy(base)
    for k, v in kv.items(): setattr(p, k, v)
    zc = z.cctx(a.level, producer=orc.producer_addr, state=B.C.addressof(p), ext_repcodes=a.ext_rep)
    c, _ = z.compress_chunks(zc, data, 131072); z.free(zc)
    print("%-16s %9d  ours/sw = %.4f" % (name, c, c / sw))
