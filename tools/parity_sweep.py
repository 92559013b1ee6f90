#!/usr/bin/env python3
"""Large parity sweep on the GPU box: N blocks of the system corpus per level, GPU sequences vs the oracle's (threads over
the oracle .so), whole blocks and — every fourth block — its last-segment work item.  usage: parity_sweep.py [levels...]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import qz_bind as B, qz_corpus as K

levels = [int(a, 0) for a in sys.argv[1:]] or [1, 2, 3, 5, 6, 9, 12, 0x101]
nb = int(os.environ.get("QZ_BLOCKS", "1024"))
plug, orc = B.Plugin(B.PLUGIN_SO), B.Oracle()
data = K.system_corpus(nb * 131072)[0]
blocks = [data[o:o + 131072] for o in range(0, len(data), 131072)]
items = blocks + [b for b in blocks[::4]]
froms = [0] * len(blocks) + [98304] * len(blocks[::4])
bad = 0
for level in levels:
    t0 = time.time()
    counts, seqs, stride = plug.find_batch(items, level, parse_from=froms)
    a = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)
    def check(i):
        n, want = orc.find(orc.profile(level, len(items[i])), items[i], cap=stride, parse_from=froms[i])
        if counts[i] != n:
            return 1
        w = np.frombuffer(want, dtype=np.uint32).reshape(-1, 4)[:n, :3]
        return 0 if np.array_equal(a[i * stride:i * stride + n, :3], w) else 1
    with ThreadPoolExecutor(max_workers=16) as ex:
        wrong = sum(ex.map(check, range(len(items))))
    bad += wrong
    print("level %#x: %d items (%d segment items), %d differ from the oracle, %.1f s" % (level, len(items), len(items) - len(blocks), wrong, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
