#!/usr/bin/env python3
"""What a segment work item costs (GPU box, under rocprofv3 --kernel-trace --stats): three launches of N items each —
whole 128 KiB blocks, the same blocks as segment items parsing the last 32 KiB (192 history tiles + 64 parsed), and the last
32 KiB alone as blocks of their own (64 tiles).  (history tile cost) = (segment - alone) / 192."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B, qz_corpus as K

n = int(os.environ.get("QZ_BLOCKS", "256"))
level = int(os.environ.get("QZ_LEVEL", "1"), 0)
plug = B.Plugin(B.PLUGIN_SO)
data = K.system_corpus(n * 131072)[0]
blocks = [data[o:o + 131072] for o in range(0, len(data), 131072)]
for rep in range(2):
    plug.find_batch(blocks, level)
    plug.find_batch(blocks, level, parse_from=[98304] * n)
    plug.find_batch([b[98304:] for b in blocks], level)
print("done: launches 1-3 are the warm-up, 4 = whole blocks, 5 = segment items (parse from 96 KiB), 6 = the last 32 KiB alone")
