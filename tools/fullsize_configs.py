#!/usr/bin/env python3
"""GPU box: BASELINE.json's configs 3, 4 and 5 AT THEIR STATED SIZES on the one GPU there is (round-5 verdict, weak 1: "configs 3-5 are
exercised in shape at 256 MiB, never at the stated 1 GB / 16 GiB / 64 GiB").  The corpora are stand-ins (no enwik9 / Silesia in the image; pass
--enwik9 PATH when there is one); sizes, levels, block sizes and framing are the configs'.  Every frame produced is decoded again and compared
with its input (the reference's own criterion, /root/reference/test/benchmark.c:329-339); compressed sizes are set against libzstd's own
match-finder on the same bytes (north star: within 2 %).

  config 3   level 6, 128 KiB blocks, 1 GB (10^9 bytes), through the batch front-end; software level 6 over the same bytes
  config 4   level 12, 32 KiB blocks, 16 GiB of synthetic web-log lines, 1 GiB at a time (the 8-GPU shard of the config on one GPU, one after the other)
  config 5   level 3, ZSTD_compress2 over 4 MiB frames (32 producer calls per frame) with 4 MiB announced ahead, 64 GiB of a mixed-entropy
             stream = 16 threads x 16 passes x 256 MiB through qat-zstd-plugin_amd/test/benchmark (each thread its own CCtx + state)

usage: python tools/fullsize_configs.py [--configs 3,4,5] [--out FILE] [--enwik9 PATH] [--scale 1.0]    (--scale 0.01: a smoke run)
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import qz_bind as B  # noqa: E402
import qz_corpus as K  # noqa: E402


def threads():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = float(q) / float(per) if q != "max" else float(os.cpu_count() or 1)
    except Exception:  # noqa: BLE001
        quota = float(os.cpu_count() or 1)
    return max(1, min(int(quota), len(os.sched_getaffinity(0)), 128))


class FrontRun:
    """ONE front (a pool of CCtx threads with their states, streams and pinned buffers) used for every pass of a config: the device layer's start-up and
    the buffers' first allocation are not a config's time — the first pass over the first bytes is run twice, the first time untimed"""

    def __init__(self, front, chunk: int, level: int, nthreads: int, use_producer: int):
        self.F, self.chunk, self.use_producer = front.lib, chunk, use_producer
        prm = B.FrontParams(nthreads, level, chunk, 0, 0, use_producer)
        self.f = self.F.QZSTD_createFront(C.byref(prm))
        assert self.f, "QZSTD_createFront failed"
        self.stride = self.F.QZSTD_frontFrameStride(self.f)
        self.dst, self.sizes, self.cap = None, None, 0
        self.z = B.Zstd()
        self.warm = False

    def run(self, data: bytes, verify: bool):
        """-> (seconds of QZSTD_frontCompress, compressed bytes, every frame decoded to its input)"""
        F, chunk, stride = self.F, self.chunk, self.stride
        n = (len(data) + chunk - 1) // chunk
        if n > self.cap:
            self.dst, self.sizes, self.cap = C.create_string_buffer(n * stride), (C.c_size_t * n)(), n
        if not self.warm:
            assert F.QZSTD_frontCompress(self.f, data, len(data), self.dst, len(self.dst), self.sizes) == n
            self.warm = True
        t0 = time.perf_counter()
        got = F.QZSTD_frontCompress(self.f, data, len(data), self.dst, len(self.dst), self.sizes)
        dt = time.perf_counter() - t0
        assert got == n, "QZSTD_frontCompress returned %d of %d frames" % (got, n)
        ok = True
        if verify:
            back = C.create_string_buffer(chunk)
            for c in range(n):
                want = data[c * chunk:(c + 1) * chunk]
                r = self.z.lib.ZSTD_decompress(back, chunk, C.byref(self.dst, c * stride), self.sizes[c])
                if self.z.is_error(r) or r != len(want) or back.raw[:r] != want:
                    ok = False
                    break
        return dt, sum(self.sizes[c] for c in range(n)), ok

    def close(self):
        st, fs = (C.c_ulong * 2)(), (C.c_ulong * 8)()
        self.F.QZSTD_frontStats(self.f, st)
        self.F.QZSTD_frontFailStats(self.f, fs)
        self.F.QZSTD_freeFront(self.f)
        return list(st), list(fs)


def config3(a, front, nt):
    size = int(1_000_000_000 * a.scale)
    if a.enwik9 and os.path.isfile(a.enwik9):
        raw = open(a.enwik9, "rb").read(size)
        prov = "file:" + a.enwik9
    else:
        unit = K.system_corpus(64 * K.MiB)[0] + K.text(3, 32 * K.MiB) + K.weblog(5, 16 * K.MiB)
        raw = (unit * (-(-size // len(unit))))[:size]
        prov = "stand-in for enwik9: system corpus (64 MiB) + Zipf text (32 MiB) + web-log lines (16 MiB), repeated to size"
    g = FrontRun(front, 131072, 6, nt, 1)
    t, csize, ok = g.run(raw, True)
    st, fs = g.close()
    w = FrontRun(front, 131072, 6, nt, 0)
    t_sw, csize_sw, _ = w.run(raw, False)
    w.close()
    st[0] //= 2  # (the untimed first pass was served from announcements as well)
    return {"config": "level-6, 128 KiB blocks, %d bytes (BASELINE configs[2]: enwik9, 1 GB) on 1 x MI355X" % size, "corpus": prov, "threads": nt,
            "MBps": round(size / t / 1e6, 1), "csize": csize, "ratio": round(size / csize, 4),
            "software_level6_same_bytes": {"MBps": round(size / t_sw / 1e6, 1), "csize": csize_sw},
            "csize_vs_software": round(csize / csize_sw, 4), "within_2pct": csize <= 1.02 * csize_sw,
            "blocks_from_announcements": st[0], "blocks_per_block_path": st[1], "producer_errors": fs[0],
            "every_frame_decodes_to_its_input": ok}


def config4(a, front, nt):
    total = int((16 << 30) * a.scale)
    gib = min(1 << 30, total)
    units = [K.weblog(40 + s, 64 * K.MiB) for s in range(2)]  # (the generator is Python: two units of 64 MiB, interleaved and rotated per GiB)
    done = csize = csize_sw = 0
    t_gpu = t_sw = 0.0
    ok = True
    h = hashlib.sha256()
    k = 0
    g, w = FrontRun(front, 32768, 12, nt, 1), FrontRun(front, 32768, 12, nt, 0)
    while done < total:
        n = min(gib, total - done)
        rot = (k * 7919 * 32768) % len(units[0])
        buf = b"".join((units[(k + j) % 2][rot:] + units[(k + j) % 2][:rot]) for j in range(-(-n // len(units[0]))))[:n]
        t, c, good = g.run(buf, True)
        t_gpu += t; csize += c; ok = ok and good
        if k < 2:  # software level 12 runs at 1.4 GB/s on these cores: the first two GiB are the ratio's denominator
            ts, cs, _ = w.run(buf, False)
            t_sw += ts; csize_sw += cs
            sw_bytes = done + n
            gpu_csize_on_sw_bytes = csize
        h.update(hashlib.sha256(buf[:1 << 20]).digest())
        done += n
        k += 1
    (ann, perblk), fsg = g.close()
    w.close()
    errs = fsg[0]
    return {"config": "level-12, 32 KiB blocks, %d bytes of synthetic web-log lines (BASELINE configs[3]: 16 GiB sharded over 8 GPUs — here the whole batch on ONE GPU, 1 GiB at a time)" % total,
            "threads": nt, "MBps": round(total / t_gpu / 1e6, 1), "seconds_in_QZSTD_frontCompress": round(t_gpu, 2), "csize": csize, "ratio": round(total / csize, 4),
            "software_level12_first_%d_bytes" % sw_bytes: {"MBps": round(sw_bytes / t_sw / 1e6, 1), "csize": csize_sw},
            "csize_vs_software_on_those_bytes": round(gpu_csize_on_sw_bytes / csize_sw, 4), "within_2pct": gpu_csize_on_sw_bytes <= 1.02 * csize_sw,
            "blocks_from_announcements": ann, "blocks_per_block_path": perblk, "producer_errors": errs,
            "every_frame_decodes_to_its_input": ok, "corpus_fingerprint": h.hexdigest()[:16]}


def config5(a, nt):
    per_thread = max(4 << 20, int((256 << 20) * min(1.0, a.scale * 16)))
    loops = max(1, -(-int((64 << 30) * a.scale) // (nt * per_thread)))  # 64 GiB in all: 16 passes of 256 MiB per thread on 16 threads
    data = K.mixed_entropy(5, per_thread)
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "benchmark", "ZSTDLIB=" + B.find_libzstd()], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import re
    res = {}
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(data)
        name = f.name
    try:
        for key, mode, extra, lp in (("plugin", 1, ["-H4"], loops), ("software", 0, [], max(1, loops // 4))):
            cmd = [os.path.join(tdir, "benchmark"), "-m%d" % mode, "-t%d" % nt, "-l%d" % lp, "-c4M", "-L3", "-S1", "-P1"] + extra + [name]
            t0 = time.perf_counter()
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
            wall = time.perf_counter() - t0
            agg = re.search(r"([0-9.]+) MB/s by the wall clock of the compression phase \(([0-9.]+) s\)", out.stderr)
            first = re.search(r"Compression: (\d+) -> (\d+) ", out.stderr)
            er = re.search(r"[Pp]roducer errors: (\d+)", out.stderr)
            res[key] = {"returncode": out.returncode, "MBps_wall": float(agg.group(1)) if agg else None, "compression_phase_s": float(agg.group(2)) if agg else None,
                        "bytes": nt * lp * per_thread, "csize_per_thread_buffer": int(first.group(2)) if first else None,
                        "round_trips_PASS": out.stderr.count("PASS"), "threads": nt, "producer_errors": int(er.group(1)) if er else None,
                        "tool": " ".join(cmd[:-1]), "wall_s": round(wall, 1)}
            if out.returncode != 0:
                res[key]["stderr_tail"] = out.stderr[-400:]
    finally:
        os.unlink(name)
    p, s = res.get("plugin", {}), res.get("software", {})
    return {"config": "level-3, ZSTD_compress2 per 4 MiB frame (blocks of 128 KiB: -S1), 4 MiB announced ahead, %d bytes of a mixed-entropy stream = %d threads x %d passes x %d MiB "
                      "(BASELINE configs[4]: 64 GiB, ZSTD_compressStream2 chunked 4 MiB frames)" % (p.get("bytes", 0), nt, loops, per_thread >> 20),
            "plugin": p, "software": s,
            "csize_vs_software": round(p["csize_per_thread_buffer"] / s["csize_per_thread_buffer"], 4) if p.get("csize_per_thread_buffer") and s.get("csize_per_thread_buffer") else None,
            "note": "the one config where the north star's 2 % cannot hold: software matches across the whole 4 MiB frame, the producer contract parses every 128 KiB block "
                    "without history (/root/reference/src/qatseqprod.h:103-105); on this corpus the incompressible and constant segments dominate"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3,4,5")
    ap.add_argument("--out", default=None)
    ap.add_argument("--enwik9", default=None)
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    B.Zstd()
    plug = B.Plugin()
    assert plug.lib.qzstd_hip_device_count() > 0, plug.err()
    front = B.Front()
    nt = threads()
    out = {"threads": nt, "libzstd": B.Zstd().version(), "scale": a.scale}
    for c in a.configs.split(","):
        t0 = time.perf_counter()
        try:
            out["config%s" % c] = {"3": lambda: config3(a, front, nt + max(1, nt // 16)), "4": lambda: config4(a, front, nt + max(1, nt // 16)), "5": lambda: config5(a, nt)}[c]()
        except Exception as e:  # noqa: BLE001
            out["config%s" % c] = {"error": repr(e)[:400]}
        out["config%s" % c]["wall_s_including_verification"] = round(time.perf_counter() - t0, 1)
        print("# config %s: %s" % (c, json.dumps(out["config%s" % c])), file=sys.stderr, flush=True)
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
