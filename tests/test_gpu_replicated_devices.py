"""GPU: the LIBRARY'S OWN N > 1 path on real HIP (round-5 verdict, "What's missing" 1 / "Next round" 6).

`north_star`: "the batch shards across the 8 GPUs of one node with per-GPU HIP streams and host-side gather (no RCCL)"; the reference's
analogue is the instance spread of /root/reference/src/qatseqprod.c:601-630, :1156-1162.  Until now QZSTD_hintSource's cut of an
announcement into per-GPU block ranges, the placement of states over the GPUs and one resident service per device had only run against
tests/mock/mock_hip.c: the GPU boxes have one MI355X.  QZSTD_HIP_REPLICATE_DEVICES=2 (test only, csrc/qzstd_kernels.hip: probe_devices)
lists the physical GPU twice, so the library sees two LOGICAL devices — own streams, own pinned buffers, own batches, own resident
service each — and the whole path runs on real HIP.  Everything has to come out as with one device: frames byte-identical to libzstd's
frames from the ORACLE's sequences, no producer error, and BOTH devices have to have served blocks (QZSTD_deviceStats).

The variable is read once per process (device enumeration), so every case runs in a child process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes as C, json, os, sys, threading
sys.path.insert(0, os.path.join(%(root)r, "tools"))
import qz_bind as B, qz_corpus as K
plug = B.Plugin(); L = plug.lib
z = B.Zstd(); orc = B.Oracle()
out = {"devices": L.qzstd_hip_device_count()}
L.QZSTD_deviceStats.argtypes = [C.c_int, C.POINTER(C.c_ulong * 4)]

def dev_stats():
    res = []
    for d in range(L.QZSTD_deviceStats(0, None)):
        s = (C.c_ulong * 4)()
        L.QZSTD_deviceStats(d, C.byref(s))
        res.append(list(s))
    return res

def oracle_frames(data, chunk, level):
    zo = z.cctx(level, producer=orc.producer_addr, state=None, fallback=False, validate=True)
    _, fr = z.compress_chunks(zo, data, chunk)
    z.free(zo)
    return fr

mode = sys.argv[1]
if mode == "announced":
    # the batch front-end: claims of `seg` bytes are announced and every announcement is cut into one block range per (logical) GPU
    level, chunk, nblk, seg = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    data = K.by_name("system", nblk * chunk + 777, seed=level + 200)
    front = B.Front()
    frames, st, fs = front.frames(data, chunk, level, 4, segment=seg, jobs=2)
    want = oracle_frames(data, chunk, level)
    out.update(frames=len(frames), bad=[c for c in range(len(want)) if frames[c] != want[c]][:8], front_stats=st, fail=fs[0], dev=dev_stats())
elif mode == "callers":
    # unchanged callers: one CCtx + one state per thread (README.md:138 of the reference), nothing announced: the states are spread over the
    # devices, every device has its own resident service (levels 1-2, 5-12) and its own batches
    level, chunk, nthreads = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    assert L.QZSTD_startQatDevice() == 0
    datas = [K.by_name("mix", 5 * chunk + 100 * t, seed=300 + t) for t in range(nthreads)]
    res, errs = [None] * nthreads, [0] * nthreads
    def work(t):
        zt = B.Zstd(z.path)
        st = L.QZSTD_createSeqProdState()
        zc = zt.cctx(level, producer=plug.producer_addr, state=st, fallback=False, validate=True)
        _, res[t] = zt.compress_chunks(zc, datas[t], chunk)
        zt.free(zc)
        f = (C.c_ulong * 8)(); L.QZSTD_failStats(st, C.byref(f)); errs[t] = int(f[0])
        L.QZSTD_freeSeqProdState(st)
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    bad = [t for t in range(nthreads) if res[t] != oracle_frames(datas[t], chunk, level)]
    out.update(bad=bad, fail=sum(errs), dev=dev_stats())
    L.QZSTD_stopQatDevice()
print("RESULT " + json.dumps(out))
'''


def run_child(args, env_extra=None, timeout=600):
    env = dict(os.environ, QZSTD_HIP_REPLICATE_DEVICES="2", **(env_extra or {}))
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}] + [str(a) for a in args], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("RESULT ")]
    assert line, (out.stdout + out.stderr)[-2000:]
    return json.loads(line[-1][7:])


@pytest.mark.parametrize("level,chunk,nblk,seg", [(1, 131072, 128, 16 << 20), (6, 131072, 64, 4 << 20), (12, 32768, 256, 2 << 20)])
def test_announcement_is_cut_into_one_range_per_device_and_both_serve(gpu_plugin, level, chunk, nblk, seg):
    r = run_child(["announced", level, chunk, nblk, seg])
    assert r["devices"] == 2, r
    assert not r["bad"], "frames %s differ from libzstd + oracle with two logical devices" % r["bad"]
    assert r["fail"] == 0, r
    assert r["front_stats"][1] == 0 and r["front_stats"][0] == 2 * r["frames"], "not every block came from an announcement: %s" % r
    ann = [d[0] for d in r["dev"]]
    assert len(ann) == 2 and min(ann) > 0, "one of the two devices served no announced block: %s" % r["dev"]
    assert sum(ann) == 2 * r["frames"], r  # every block of both jobs counted once, on the device whose range held it


def test_split_one_keeps_an_announcement_on_its_states_device(gpu_plugin):
    """QZSTD_HIP_SPLIT=1: announcements are not cut; the states (four front-end workers) are spread over the two devices instead"""
    r = run_child(["announced", 1, 131072, 96, 2 << 20], {"QZSTD_HIP_SPLIT": "1"})
    assert r["devices"] == 2 and not r["bad"] and r["fail"] == 0, r
    assert min(d[0] for d in r["dev"]) > 0, r["dev"]


@pytest.mark.parametrize("level", [1, 6])
def test_unchanged_callers_are_spread_over_both_devices_services(gpu_plugin, level):
    r = run_child(["callers", level, 131072, 6])
    assert r["devices"] == 2 and not r["bad"] and r["fail"] == 0, r
    served = [d[1] + d[2] for d in r["dev"]]  # batches + resident service
    assert min(served) > 0, "one of the two devices served no per-block request: %s" % r["dev"]
    assert sum(d[2] for d in r["dev"]) > 0, "no block went through a resident service: %s" % r["dev"]


def test_bench_product_multi_gpu_leg_runs_with_two_devices(gpu_plugin):
    """bench.py's `product_multi_gpu` leg (one process, every device: the PCIe pipeline on all of them at once, the front-end with
    QZSTD_HIP_SPLIT = 1 and = N) with gpus: 2 — the leg the driver's 8-GPU run will execute"""
    env = dict(os.environ, QZSTD_HIP_REPLICATE_DEVICES="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--product-multi-gpu", "2", "--level", "1", "--block", "131072"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("{")]
    r = json.loads(line[-1])
    assert r.get("gpus") == 2, r
    assert "GBps_input_aggregate" in r["pcie_pipeline"], r["pcie_pipeline"]
    for key in ("split_1", "split_2"):
        fe = r["frontend"][key]
        assert fe.get("roundtrip") == "PASS" and fe.get("producer_errors", {}).get("total") == 0, (key, fe)
        assert fe["blocks_per_block_path"] == 0, (key, fe)
        per = fe["blocks_per_gpu_announced_batched_service"]
        assert "gpu0" in per and "gpu1" in per, per
        assert all(int(x.split()[1].split("/")[0]) > 0 for x in per.split(", ")), "a device served nothing: %s" % per
