#!/usr/bin/env python3
"""GPU box: one request through the resident service by hand (C ABI), with the service's counters printed while waiting.
usage: gpurun -- python tools/svc_probe.py [level]   (env QZSTD_HIP_SERVICE_WORKERS, QZSTD_HIP_SERVICE_IDLE_US)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B, qz_corpus as K


class Req(C.Structure):
    _fields_ = [("hSrc", C.c_void_p), ("dSrc", C.c_void_p), ("hSeqs", C.c_void_p), ("hCount", C.c_void_p),
                ("srcLen", C.c_uint32), ("itemBytes", C.c_uint32), ("nItems", C.c_uint32), ("seqCapPerItem", C.c_uint32),
                ("slot", C.c_uint32), ("epoch", C.c_uint32)]


def main():
    level = int(sys.argv[1], 0) if len(sys.argv) > 1 else 1
    plug, orc = B.Plugin(), B.Oracle()
    L = plug.lib
    L.qzstd_hip_host_alloc_coherent.restype = C.c_void_p
    L.qzstd_hip_host_alloc_coherent.argtypes = [C.c_size_t]
    L.qzstd_hip_service_submit.argtypes = [C.c_int, C.c_int, C.POINTER(Req)]
    assert L.qzstd_hip_device_count() >= 1, plug.err()
    n, item, cap = 131072, 4096, 1371
    nit = n // item
    hsrc = L.qzstd_hip_host_alloc_coherent(n + 64)
    hseq = L.qzstd_hip_host_alloc_coherent(nit * cap * 16)
    hcnt = L.qzstd_hip_host_alloc_coherent(nit * 4)
    dsrc = L.qzstd_hip_malloc(0, n + 64)
    data = K.system_corpus(8 * n)[0]
    print("hsrc %#x..%#x hseq %#x..%#x hcnt %#x dsrc %#x..%#x" % (hsrc, hsrc + n + 64, hseq, hseq + nit * cap * 16, hcnt, dsrc, dsrc + n + 64), flush=True)
    if os.environ.get("NITEMS"):
        nit = int(os.environ["NITEMS"]); item = n // nit; cap = 32 * 1371 // nit
    info, dbg = (C.c_ulong * 8)(), (C.c_ulong * 8)()
    cnt = (C.c_uint32 * nit).from_address(hcnt)
    for rep in range(int(os.environ.get("REPS", "6"))):
        blk = data[rep * n:(rep + 1) * n]
        C.memmove(hsrc, blk, n)
        for k in range(nit):
            cnt[k] = 0
        rq = Req(hsrc, dsrc, hseq, hcnt, n, item, nit, cap, 5, rep + 1)
        t0 = time.perf_counter()
        rc = L.qzstd_hip_service_submit(0, level, C.byref(rq))
        t1 = time.perf_counter()
        deadline = t0 + 1.0
        nextp = t0 + 0.1
        while time.perf_counter() < deadline and not all(cnt[k] for k in range(nit)):
            if time.perf_counter() > nextp:
                L.qzstd_hip_service_info(0, C.byref(info)); L.qzstd_hip_service_debug(0, C.byref(dbg))
                print("   waiting: arrived %d/%d info %s dbg %s" % (sum(1 for k in range(nit) if cnt[k]), nit, list(info), list(dbg)), flush=True)
                nextp += 0.2
        t2 = time.perf_counter()
        got = [cnt[k] for k in range(nit)]
        ok = all(got)
        # check against the oracle, item by item
        bad = 0
        if ok:
            pf = orc.profile(level, n)
            seqs = (B.Sequence * (nit * cap)).from_address(hseq)
            for k in range(nit):
                wn, want = orc.find(pf, blk[:(k + 1) * item], cap=cap, parse_from=k * item)
                if wn != got[k] or any((seqs[k * cap + i].offset, seqs[k * cap + i].litLength, seqs[k * cap + i].matchLength) !=
                                       (want[i].offset, want[i].litLength, want[i].matchLength) for i in range(wn)):
                    bad += 1
        L.qzstd_hip_service_info(0, C.byref(info)); L.qzstd_hip_service_debug(0, C.byref(dbg))
        print("request %d: submit rc %d in %.1f us, all counts after %.1f us, complete %s, items differing from the oracle %d; info %s dbg %s"
              % (rep, rc, (t1 - t0) * 1e6, (t2 - t0) * 1e6, ok, bad, list(info), list(dbg)), flush=True)
        if not ok:
            break
    print("stop:", L.qzstd_hip_service_stop(0), flush=True)
    L.qzstd_hip_service_info(0, C.byref(info))
    print("after stop info", list(info))


main()
