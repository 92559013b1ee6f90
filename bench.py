#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native ZSTD block-level sequence producer.

A "step" is one pass of the hot path (the HIP match-finder behind qatSequenceProducer,
called through the C ABI qzstd_hip_find_sequences) over one batch of blocks that is
already resident in HBM: BASELINE.json configs[1] = level 1, 128 KiB blocks, 1 GiB
Silesia-like batch (8192 blocks) per GPU.  One process per GPU (torch.distributed.run);
blocks are independent, so ranks share nothing on the data path (weak scaling, no
collective besides the timing barrier / MAX).

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit, roofline
(HBM-bound integer kernel: algorithmic bytes = block bytes read + 16 B per sequence
written), cpu_baseline (the CPU oracle timed on a bounded sample), plus the
north-star's own CPU baseline (libzstd's internal match-finder, plugin unregistered)
and the compression ratio of the produced sequences vs software zstd.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import qz_bind as B  # noqa: E402
import qz_corpus as K  # noqa: E402
import qz_shard as S  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def profiled_traffic(blocks: int, block: int):
    """HBM bytes per launch from the newest committed PMC summary (profiles/rNN_pmc_summary.txt,
    produced by tools/prof_pmc.sh with separate --pmc passes on this same workload).  Corrected as
    MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are KiB, and on gfx950 FETCH_SIZE
    reports half of a 16 B/lane coalesced stream.  None when no summary matches the workload."""
    import glob
    import re
    if (blocks, block) != (8192, 131072):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
    if not files:
        return None, None
    txt = open(files[-1]).read()
    f = re.search(r"FETCH_SIZE\s+per-launch\s+([0-9.]+)", txt)
    w = re.search(r"WRITE_SIZE\s+per-launch\s+([0-9.]+)", txt)
    if not (f and w):
        return None, None
    return int((2.0 * float(f.group(1)) + float(w.group(1))) * 1024), os.path.basename(files[-1])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--level", type=lambda v: int(v, 0), default=1, help="1..12, optionally | 0x100 (repeat-offset aware)")
    ap.add_argument("--block", type=int, default=131072)
    ap.add_argument("--blocks", type=int, default=8192, help="blocks per GPU per step (8192 x 128 KiB = 1 GiB)")
    ap.add_argument("--corpus", default="system", help="system | text | mix | weblog | mixed_entropy | /path/to/file")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / end-to-end legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def load_corpus(name: str, size: int) -> tuple[bytes, str]:
    if os.path.isfile(name):
        with open(name, "rb") as f:
            raw = f.read()
        reps = -(-size // len(raw))
        return (raw * reps)[:size], "file:%s(%d) repeated" % (name, len(raw))
    if name == "system":
        return K.system_corpus(size)
    seed = {"text": 1, "mix": 2, "weblog": 4, "mixed_entropy": 5}.get(name, 1)
    unit = min(size, 64 * K.MiB)
    raw = K.by_name(name, unit, seed)
    reps = -(-size // len(raw))
    return (raw * reps)[:size], "synthetic:%s(seed=%d,%d B) repeated" % (name, seed, unit)


# ----------------------------------------------------------------------------- CPU legs
def cpu_oracle_leg(data: bytes, block: int, level: int, seconds: float):
    """cpu_baseline: the CPU restatement of the same match-finder ("port"), 1 thread."""
    orc = B.Oracle()
    prof = orc.profile(level, block)
    cap = B.sequence_bound(block)
    out = (B.Sequence * cap)()
    done = 0
    t0 = time.perf_counter()
    o = 0
    while time.perf_counter() - t0 < seconds and o + block <= len(data):
        orc.lib.qzo_find_sequences(C.byref(prof), data[o:o + block], block, out, cap)
        o += block
        done += block
    dt = time.perf_counter() - t0
    return {"value": round(done / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "oracle/qzstd_oracle.c qzo_find_sequences, first %d blocks of the batch, 1 thread" % (done // block)}


def c_benchmark(sample: bytes, block: int, level: int, threads: int, mode: int, hint: int = 0, ext_rep: int = 0,
                loops: int = 2, env: dict | None = None):
    """run qat-zstd-plugin_amd/test/benchmark (counterpart of the reference's test/benchmark.c: T threads,
    one CCtx each, one ZSTD_compress2 per chunk, each chunk its own frame) on a sample file"""
    import re
    import subprocess
    import tempfile
    tdir = os.path.join(B.PKG_DIR, "test")
    zpath = B.find_libzstd()
    try:
        subprocess.check_call(["make", "-C", tdir, "benchmark", "ZSTDLIB=" + zpath], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            f.write(sample)
            name = f.name
        cmd = [os.path.join(tdir, "benchmark"), "-m%d" % mode, "-t%d" % threads, "-l%d" % loops, "-c%d" % block, "-L%d" % level,
               "-E%d" % ext_rep] + (["-H%d" % hint] if hint else []) + [name]
        t0 = time.perf_counter()
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
        wall = time.perf_counter() - t0
        os.unlink(name)
        agg = re.search(r"aggregate compression ([0-9.]+) MB/s", out.stderr)
        first = re.search(r"Compression: (\d+) -> (\d+) ", out.stderr)
        lat = re.search(r"P50 ([0-9.]+)\s+P75 [0-9.]+\s+P99 ([0-9.]+)", out.stderr)
        if out.returncode != 0 or not agg or not first:
            return {"error": (out.stderr or "benchmark failed")[-300:]}
        return {"MBps": float(agg.group(1)), "threads": threads, "bytes_per_thread": int(first.group(1)),
                "csize": int(first.group(2)), "ratio": round(int(first.group(1)) / max(int(first.group(2)), 1), 4),
                "latency_us_p50": float(lat.group(1)) if lat else None, "latency_us_p99": float(lat.group(2)) if lat else None,
                "tool": "qat-zstd-plugin_amd/test/benchmark " + " ".join(cmd[1:-1]), "wall_s": round(wall, 2)}
    except Exception as e:  # noqa: BLE001 - the bench line must still be printed
        return {"error": repr(e)[:300]}


# ----------------------------------------------------------------------------- main
def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    plug = B.Plugin()
    L = plug.lib
    assert L.qzstd_hip_device_count() > local, plug.err()

    block, nb, level = a.block, a.blocks, a.level
    size = block * nb
    # every rank gets its own shard: rank r starts r*size/world into the (repeated) corpus
    off = S.weak_offset(size, world, rank)
    data, prov = load_corpus(a.corpus, size + S.weak_offset(size, world, world - 1))
    shard = data[off:off + size]

    dev = torch.device("cuda", local)
    d_src = torch.empty(size + 64, dtype=torch.uint8, device=dev)
    d_src[:size].copy_(torch.frombuffer(bytearray(shard), dtype=torch.uint8))
    stride = B.sequence_bound(block)
    d_seqs = torch.empty((nb * stride, 4), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
    desc = (B.HipBlock * nb)()
    for i in range(nb):
        desc[i].srcOff = i * block
        desc[i].seqOff = i * stride
        desc[i].srcLen = block
        desc[i].seqCap = stride
    d_desc = torch.empty(C.sizeof(desc), dtype=torch.uint8, device=dev)
    d_desc.copy_(torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8))
    work = L.qzstd_hip_workspace_bytes(level, nb, block)  # levels >= 6: hash chains in device memory
    d_work = torch.empty(max(work, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()

    def step():
        rc = L.qzstd_hip_find_sequences(local, C.c_void_p(stream.cuda_stream), level, C.c_void_p(d_src.data_ptr()),
                                        C.c_void_p(d_desc.data_ptr()), nb, block, C.c_void_p(d_seqs.data_ptr()),
                                        C.c_void_p(d_cnt.data_ptr()), C.c_void_p(d_work.data_ptr()), work)
        if rc != 0:
            raise RuntimeError(plug.err())

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(a.steps):
        step()
        ev[k + 1].record(stream)  # HIP events on the launch stream: per-launch kernel time
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall = S.reduce_max_seconds(wall, dist if world > 1 else None, dev)
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(a.steps)]
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    cnt = d_cnt.cpu().numpy().astype("uint32")
    n_err = int((cnt == 0xFFFFFFFF).sum())
    seq_total = int(cnt[cnt != 0xFFFFFFFF].sum())

    if rank == 0:
        total_bytes = size * world * a.steps
        value = total_bytes / wall / 1e6
        alg_bytes = size + 16 * seq_total  # per launch: block bytes read once + 16 B per sequence written
        achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
        traffic, traffic_src = profiled_traffic(nb, block) if level == 1 else (None, None)
        out = {
            "metric": "input MB/s via ZSTD_compress2 L1 128KiB blocks @1/2/4/8 GPU; ratio vs sw zstd",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(wall / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "level-%d, %d KiB blocks, %d blocks (%.2f GiB) per GPU, sequence production "
                                   "(qzstd_hip_find_sequences = the kernel behind qatSequenceProducer), inputs resident in HBM"
                                   % (level, block >> 10, nb, size / 2 ** 30),
                       "corpus": prov[:300], "level": level, "block_bytes": block, "blocks_per_gpu": nb,
                       "parallelism": "block-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "qzstd_find_sequences_kernel", "kernel_ms_avg": round(kern_avg_ms, 3),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "input_read_frac_of_peak": round(size / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "sequences_per_block": round(seq_total / max(nb - n_err, 1), 1), "error_blocks": n_err,
        }
        if not a.no_cpu and world == 1:  # CPU legs on rank 0 at N=1 only (bench contract)
            ncpu = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_oracle_leg(shard, block, level, a.cpu_seconds)
            # north-star CPU baseline: libzstd's internal match-finder (plugin unregistered), benchmark.c shape,
            # measured by the C tool in the same run on the host cores of this box
            sample = shard[:min(len(shard), 512 * block)]
            thr = min(ncpu, 16)  # <= QZ_DEFAULT_SLOTS_PER_DEVICE: one slot per thread
            sw = c_benchmark(sample, block, level, thr, mode=0, loops=6)
            out["cpu_libzstd_sw"] = {"lib": os.path.basename(B.find_libzstd()), "host_cores": ncpu, **sw}
            # end to end through ZSTD_compress2 with the plugin registered: 4 MiB look-ahead hints (the GPU match-finds
            # segment k+1 while the thread entropy-codes segment k) ...
            e2e = c_benchmark(sample, block, level, thr, mode=1, hint=1, loops=6)
            if "csize" in e2e and "csize" in sw:
                e2e["csize_vs_sw"] = round(e2e["csize"] / sw["csize"], 4)
                e2e["ratio_within_2pct"] = e2e["csize"] <= sw["csize"] * 1.02
            out["e2e_zstd_compress2_plugin"] = e2e
            # ... the same with ZSTD_c_searchForExternalRepcodes on (-E1) and the repeat-offset aware parse ...
            rep = c_benchmark(sample, block, level, thr, mode=1, hint=16, ext_rep=1, loops=6,
                              env={"QZSTD_HIP_EXT_REPCODES": "1"})
            if "csize" in rep and "csize" in sw:
                rep["csize_vs_sw"] = round(rep["csize"] / sw["csize"], 4)
            out["e2e_zstd_compress2_plugin_repcodes"] = rep
            # ... and for unchanged callers (no hints at all): the plugin's transparent look-ahead guesses that the bytes
            # behind the current block come next (fault-safe read, verified when used); what misses goes per block
            # through the coalescer
            un = c_benchmark(sample, block, level, thr, mode=1, loops=6)
            if "csize" in un and "csize" in sw:
                un["csize_vs_sw"] = round(un["csize"] / sw["csize"], 4)
            out["e2e_zstd_compress2_plugin_unchanged_callers"] = un
            out["e2e_zstd_compress2_plugin_unchanged_callers_no_lookahead"] = c_benchmark(
                sample[:128 * block], block, level, thr, mode=1, loops=4, env={"QZSTD_HIP_LOOKAHEAD": "0"})
            # BASELINE config 3's level on the same framing: software level 6 vs the plugin (hash chains + repeat-offset
            # aware parse, -E1), a quarter of the sample
            if level == 1:
                s6 = sample[:len(sample) // 4]
                sw6 = c_benchmark(s6, block, 6, thr, mode=0, loops=2)
                p6 = c_benchmark(s6, block, 6, thr, mode=1, hint=16, ext_rep=1, loops=4, env={"QZSTD_HIP_EXT_REPCODES": "1"})
                if "csize" in p6 and "csize" in sw6:
                    p6["csize_vs_sw"] = round(p6["csize"] / sw6["csize"], 4)
                    p6["speedup_vs_sw"] = round(p6["MBps"] / max(sw6["MBps"], 1e-9), 2)
                out["level6_cpu_libzstd_sw"] = sw6
                out["level6_e2e_plugin_repcodes"] = p6
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
