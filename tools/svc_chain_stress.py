#!/usr/bin/env python3
"""GPU box: many callers at once through the resident service at a chain level, every item of every request compared with results
computed beforehand by the oracle; prints where the first differences are.  usage: gpurun -- python tools/svc_chain_stress.py [level] [threads] [reps]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B, qz_corpus as K

def main():
    level = int(sys.argv[1], 0) if len(sys.argv) > 1 else 6
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    plug, orc = B.Plugin(), B.Oracle()
    data = K.by_name("system", 6 * 131072, seed=31)
    blocks = [data[o:o + 131072] for o in range(0, len(data), 131072)]
    blocks[1] = blocks[1][:100001]
    lane0 = plug.service_lane(slot=100)
    want = []
    for b in blocks:  # expected results: the oracle, once
        r = lane0.run(b, level)
        counts, seqs, cap, item = r
        pf = orc.profile(level, len(b))
        exp = []
        for k in range(len(counts)):
            n, w = orc.find(pf, b[:min(len(b), (k + 1) * item)], cap=cap, parse_from=k * item)
            exp.append([(w[i].offset, w[i].litLength, w[i].matchLength) for i in range(n)])
        want.append(exp)
    bad = []
    lanes = [plug.service_lane(slot=t) for t in range(nthreads)]
    for ln in lanes:
        ln.run(blocks[0], level)  # (the scratch is allocated on first use: before the threads start)
    def worker(t):
        lane = lanes[t]
        for rep in range(reps):
            bi = (t + rep) % len(blocks)
            r = None
            while r is None:
                r = lane.run(blocks[bi], level)
            counts, seqs, cap, item = r
            for k, n in enumerate(counts):
                exp = want[bi][k]
                got = [(seqs[k * cap + i].offset, seqs[k * cap + i].litLength, seqs[k * cap + i].matchLength) for i in range(min(n, cap))]
                if n != len(exp) or got != exp:
                    d = next((i for i in range(min(len(got), len(exp))) if got[i] != exp[i]), min(len(got), len(exp)))
                    bad.append((t, rep, bi, k, n, len(exp), d, got[d:d + 2], exp[d:d + 2]))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    [x.start() for x in th]; [x.join() for x in th]
    print("level %#x, %d threads x %d requests: %d items differ" % (level, nthreads, reps, len(bad)))
    for b in bad[:12]:
        print("  thread %d rep %d block %d item %d: count %d (oracle %d), first difference at sequence %d: %s vs %s" % b)
    plug.lib.qzstd_hip_service_stop(0)
    for ln in lanes:
        ln.close()
    lane0.close()

main()
