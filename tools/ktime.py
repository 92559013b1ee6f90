#!/usr/bin/env python3
"""GPU box: kernel time of qzstd_hip_find_sequences at given shapes, input resident in HBM, HIP events on the launch stream
(the `roofline` leg of bench.py without everything else): A/B of library builds (QZ_PLUGIN_SO=<lib>) and of profile experiments.
usage: python tools/ktime.py [level:blockBytes:blocks:corpus ...]      default: the three BASELINE kernel shapes
prints per shape: ms per launch, ms per GiB of input, input GB/s, sequences per block, error blocks"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B  # noqa: E402
import qz_corpus as K  # noqa: E402

DEFAULT = ["1:131072:8192:system", "6:131072:2048:system", "12:32768:8192:weblog"]


def corpus(name, size):
    if name == "system":
        return K.system_corpus(size)[0]
    if name == "weblog":
        unit = K.weblog(4, 64 * K.MiB)
        return (unit * (-(-size // len(unit))))[:size]
    raw = K.by_name(name, min(size, 64 * K.MiB), 1)
    return (raw * (-(-size // len(raw))))[:size]


def main():
    import torch
    plug = B.Plugin()
    L = plug.lib
    assert L.qzstd_hip_device_count() > 0, plug.err()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    reps = int(os.environ.get("KTIME_REPS", "5"))
    for spec in (sys.argv[1:] or DEFAULT):
        lv, blk, nb, name = spec.split(":")
        lv, blk, nb = int(lv, 0), int(blk), int(nb)
        data = corpus(name, blk * nb)
        d_src = torch.empty(blk * nb + 64, dtype=torch.uint8, device=dev)
        d_src[:blk * nb].copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        host_results = os.environ.get("KTIME_HOST_RESULTS", "0") not in ("", "0")  # sequences + counts written into PINNED HOST memory, as in the product paths
        stride = 16384 if host_results else B.sequence_bound(blk)
        h_seqs = h_cnt = None
        if host_results:
            L.qzstd_hip_host_alloc.restype = C.c_void_p
            L.qzstd_hip_host_device_ptr.restype = C.c_void_p
            h_seqs, h_cnt = L.qzstd_hip_host_alloc(C.c_size_t(nb * stride * 16)), L.qzstd_hip_host_alloc(C.c_size_t(nb * 4))
            assert h_seqs and h_cnt, plug.err()
            p_seqs, p_cnt = L.qzstd_hip_host_device_ptr(C.c_void_p(h_seqs)), L.qzstd_hip_host_device_ptr(C.c_void_p(h_cnt))
        else:
            d_seqs = torch.empty((nb * stride, 4), dtype=torch.int32, device=dev)
            d_cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
            p_seqs, p_cnt = d_seqs.data_ptr(), d_cnt.data_ptr()
        packed = os.environ.get("KTIME_PACKED", "0") not in ("", "0")  # 8-byte entries (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT), as the announcements ask for
        desc = (B.HipBlock * nb)()
        for i in range(nb):
            desc[i].srcOff, desc[i].seqOff, desc[i].srcLen, desc[i].seqCap = i * blk, (i * (stride // 2) if packed else i * stride), blk, stride
            desc[i].mark = (B.MARK_COMPACT | 1) if packed else 0
        d_desc = torch.empty(C.sizeof(desc), dtype=torch.uint8, device=dev)
        d_desc.copy_(torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8))
        work = L.qzstd_hip_workspace_bytes(lv, nb, blk)
        d_work = torch.empty(max(work, 4), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def go():
            rc = L.qzstd_hip_find_sequences(0, C.c_void_p(stream.cuda_stream), lv, C.c_void_p(d_src.data_ptr()), C.c_void_p(d_desc.data_ptr()), nb, blk,
                                            C.c_void_p(p_seqs), C.c_void_p(p_cnt), C.c_void_p(d_work.data_ptr()), work)
            assert rc == 0, plug.err()
        go()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record(stream)
        for k in range(reps):
            go()
            ev[k + 1].record(stream)
        torch.cuda.synchronize()
        ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))
        med = ms[len(ms) // 2]
        if host_results:
            import numpy as np
            c = np.ctypeslib.as_array((C.c_uint32 * nb).from_address(h_cnt)).copy()
        else:
            c = d_cnt.cpu().numpy().astype("uint32")
        ok = c[c != 0xFFFFFFFF]
        # a position-weighted checksum of the counts: two builds that should produce the same sequences print the same number
        chk = int((ok.astype("uint64") * (1 + (ok.size and (__import__("numpy").arange(ok.size, dtype="uint64") % 251)))).sum()) if ok.size else 0
        occ = L.qzstd_hip_occupancy(0, lv) if hasattr(L, "qzstd_hip_occupancy") else -1
        print("[%d WG/CU]%s level %#x block %d x %d %s: %.3f ms (min %.3f) = %.1f ms/GiB = %.2f GB/s in; %.1f seq/block, %d error blocks, counts checksum %d"
              % (occ, (" results in pinned host memory," if host_results else "") + (" PACKED entries," if packed else ""), lv, blk, nb, name, med, ms[0], med * (1 << 30) / (blk * nb), blk * nb / med / 1e6, float(ok.mean()) if ok.size else 0.0,
                 int((c == 0xFFFFFFFF).sum()), chk), flush=True)
        if host_results:
            L.qzstd_hip_host_free(C.c_void_p(h_seqs)); L.qzstd_hip_host_free(C.c_void_p(h_cnt))
        else:
            del d_seqs, d_cnt
        del d_src, d_desc, d_work
        torch.cuda.empty_cache()


main()
