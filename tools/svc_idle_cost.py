#!/usr/bin/env python3
"""GPU box: what does a RESIDENT service cost while nobody asks for anything?  The one-wave dispatcher polls a 64-byte request slot
and the quit word in pinned HOST memory over PCIe (qzstd_service_dispatcher, csrc/qzstd_kernels.hip); the worker workgroups poll their
queue entries in DEVICE memory.  With the idle exit pushed out (QZSTD_HIP_SERVICE_IDLE_US) one request makes the service resident,
then the dispatcher's own poll counter (qzstd_hip_service_debug [0]) is read twice, a second apart: polls per second x 72 bytes =
the PCIe read traffic of an idle dispatcher; on an 8-GPU node there are eight of them.  (By default an idle service leaves after
20 ms, so this cost is only paid between requests that are less than 20 ms apart.)
Also prints where the GPU hangs (host NUMA node) and where a node-placed pinned buffer really landed.
usage: gpurun -- python tools/svc_idle_cost.py"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("QZSTD_HIP_SERVICE_IDLE_US", "5000000")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qz_bind as B  # noqa: E402
import qz_corpus as K  # noqa: E402


def main():
    plug = B.Plugin()
    L = plug.lib
    n = L.qzstd_hip_device_count()
    assert n > 0, plug.err()
    L.qzstd_hip_host_alloc_on_node.restype = C.c_void_p
    L.qzstd_hip_host_alloc_on_node.argtypes = [C.c_size_t, C.c_int, C.c_int]
    L.qzstd_hip_host_node_of.argtypes = [C.c_void_p]
    for d in range(n):
        node = L.qzstd_hip_device_numa_node(d)
        p = L.qzstd_hip_host_alloc_on_node(1 << 20, node, 1)
        C.memset(p, 1, 1 << 20)
        print("device %d: host NUMA node %d; a 1 MiB pinned buffer asked for on that node sits on node %d" % (d, node, L.qzstd_hip_host_node_of(p)))
        L.qzstd_hip_host_free(p)
    lane = plug.service_lane(slot=3)
    blk = K.by_name("system", 131072, seed=2)
    assert lane.run(blk, 1) is not None
    dbg = (C.c_ulong * 8)()
    time.sleep(0.2)
    L.qzstd_hip_service_debug(0, C.byref(dbg)); p0, t0 = dbg[0], time.perf_counter()
    time.sleep(1.0)
    L.qzstd_hip_service_debug(0, C.byref(dbg)); p1, t1 = dbg[0], time.perf_counter()
    rate = (p1 - p0) / (t1 - t0)
    print("idle dispatcher: %.0f polls/s of the host ring = %.1f MB/s of PCIe reads (72 B per poll: 8 granules + the quit word), %.2f us per poll"
          % (rate, rate * 72 / 1e6, 1e6 / max(rate, 1)))
    # and what a request costs right after: the latency of one block while the service is resident
    ts = []
    for _ in range(20):
        t = time.perf_counter()
        assert lane.run(blk, 1) is not None
        ts.append(time.perf_counter() - t)
    ts.sort()
    print("one 128 KiB level-1 block through the resident service from Python: median %.0f us, best %.0f us" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6))
    L.qzstd_hip_service_stop(0)
    lane.close()


main()
