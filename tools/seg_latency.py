#!/usr/bin/env python3
"""GPU box: how long ONE 128 KiB block takes on the launch path when it is cut into work items of 128 KiB (one item) down to
one 4 KiB segment each (32 items), kernel time by HIP events and launch-to-sync wall time; the last item (all the history in
front of it) and the first item (none) also alone.
usage: gpurun -- python tools/seg_latency.py [level ...]"""
import ctypes as C, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B, qz_corpus as K
import torch

def main():
    only = os.environ.get("QZ_ONLY_LAST")
    levels = [int(a, 0) for a in sys.argv[1:]] or [1]
    plug = B.Plugin(); L = plug.lib
    data = K.system_corpus(64 * 131072)[0]
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    for level in levels:
        for seglog in (17, 15, 14, 13, 12):
            seg = 1 << seglog
            res = {}
            for which in ("all", "last", "first"):
                ks, ws = [], []
                for rep in range(24):
                    blk = data[(rep % 48) * 131072:(rep % 48 + 1) * 131072]
                    n = len(blk)
                    nseg = max(1, n // seg) if seglog < 17 else 1
                    items = list(range(nseg)) if which == "all" else ([nseg - 1] if which == "last" else [0])
                    cap = B.sequence_bound(min(seg, n))
                    desc = (B.HipBlock * len(items))()
                    for j, s in enumerate(items):
                        desc[j].srcOff, desc[j].seqOff = 0, j * cap
                        desc[j].srcLen = min(n, (s + 1) * seg) if nseg > 1 else n
                        desc[j].seqCap = cap
                        desc[j].parseFrom = s * seg if nseg > 1 else 0
                    d_src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
                    d_src[:n].copy_(torch.frombuffer(bytearray(blk), dtype=torch.uint8))
                    d_desc = torch.empty(C.sizeof(desc), dtype=torch.uint8, device=dev)
                    d_desc.copy_(torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8))
                    d_seqs = torch.empty((len(items) * cap, 4), dtype=torch.int32, device=dev)
                    d_cnt = torch.zeros(len(items), dtype=torch.int32, device=dev)
                    wk = L.qzstd_hip_workspace_bytes(level, len(items), n)
                    d_work = torch.empty(max(wk, 4), dtype=torch.uint8, device=dev)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0 = time.perf_counter()
                    e0.record(st)
                    rc = L.qzstd_hip_find_sequences(0, C.c_void_p(st.cuda_stream), level, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_desc.data_ptr()),
                                                    len(items), n, C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_cnt.data_ptr()), C.c_void_p(d_work.data_ptr()), wk)
                    e1.record(st)
                    assert rc == 0, plug.err()
                    torch.cuda.synchronize()
                    ws.append((time.perf_counter() - t0) * 1e6)
                    ks.append(e0.elapsed_time(e1) * 1e3)
                    c = d_cnt.cpu().numpy().astype("uint32")
                    assert (c != 0xFFFFFFFF).all() and (c != 0).all(), c
                res[which] = (statistics.median(ks[4:]), statistics.median(ws[4:]))
            print("level %s item bytes 2^%2d (%3d items): kernel %7.1f us (wall %7.1f) | last item alone %7.1f | first item alone %7.1f" %
                  (hex(level), seglog, max(1, 131072 // seg) if seglog < 17 else 1, res["all"][0], res["all"][1], res["last"][0], res["first"][0]), flush=True)

main()
