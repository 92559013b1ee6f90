#!/usr/bin/env python3
"""GPU box: the box's own ceilings, measured raw ("the box's own number via hipMemcpy D2D should be recorded next to
the peak", "secondary ceilings to report: PCIe H2D + D2H per GPU").  No kernel of this repo runs here — only the runtime's copies:

  h2d / d2h        pinned host <-> device, `hipMemcpyAsync` (torch non_blocking copies of pinned tensors), 256 MiB per copy, `streams`
                   copies in flight on separate streams, HIP events around the whole batch
  both             the same H2D and D2H batches at the same time (what the announcement path does: input in, sequences out)
  d2d              device -> device copy of 1 GiB: read + write, so the HBM traffic is twice the copied bytes
  pcie link        generation / width from sysfs (`current_link_speed`, `current_link_width` of the amdgpu device and of the bridge above it)

usage: python tools/box_ceilings.py [--mib 256] [--streams 3] [--reps 6] [--out FILE]     prints one JSON object
The reference's analogue of what is being bounded here: its staging copies into DMA memory and back
(/root/reference/src/qatseqprod.c:216-246, :1222-1227)."""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys


def pcie_links():
    """[{bdf, speed, width, max_speed, max_width, bridge: {...}}] of the amdgpu devices, from sysfs (no lspci in the image)"""
    out = []
    for dev in sorted(glob.glob("/sys/bus/pci/drivers/amdgpu/0000:*")):
        def rd(p, name):
            try:
                return open(os.path.join(p, name)).read().strip()
            except OSError:
                return None
        ent = {"bdf": os.path.basename(dev)}
        for k in ("current_link_speed", "current_link_width", "max_link_speed", "max_link_width"):
            ent[k] = rd(dev, k)
        # the link that counts is the narrowest one on the way to the root port: walk up
        path, hops = os.path.realpath(dev), []
        for _ in range(6):
            path = os.path.dirname(path)
            if not os.path.isfile(os.path.join(path, "current_link_speed")):
                break
            hops.append({"bdf": os.path.basename(path), "speed": rd(path, "current_link_speed"), "width": rd(path, "current_link_width")})
        ent["upstream"] = hops
        out.append(ent)
    return out


def gbps_of_link(speed: str | None, width: str | None):
    """one-direction payload ceiling of a PCIe link in GB/s (128b/130b coding from Gen3 on; protocol overhead not taken off)"""
    try:
        gts = float(speed.split()[0])
        w = int(width)
    except Exception:  # noqa: BLE001
        return None
    enc = 128.0 / 130.0 if gts >= 8.0 else 0.8
    return round(gts * enc * w / 8.0, 1)


def measure(mib: int = 256, streams: int = 3, reps: int = 6, device: int = 0):
    import torch
    dev = torch.device("cuda", device)
    n = mib << 20
    res = {"copy_MiB": mib, "copies_in_flight": streams, "reps": reps, "device_name": torch.cuda.get_device_name(device)}
    hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(streams)]   # H2D sources
    hd = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(streams)]   # D2H destinations
    for t in hs:
        t.random_(0, 256)
    da = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(streams)]
    db = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(streams)]
    st_in = [torch.cuda.Stream(dev) for _ in range(streams)]
    st_out = [torch.cuda.Stream(dev) for _ in range(streams)]

    def timed(fn, nbytes):
        """best and median GB/s over `reps` batches; a batch = fn() queued on its streams, timed by the host clock between two device-wide syncs
        (several streams: no single stream's events bracket the batch)"""
        import time
        rates = []
        for _ in range(reps + 1):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            rates.append(nbytes / (time.perf_counter() - t0) / 1e9)
        rates = sorted(rates[1:])  # (the first batch warms the path up)
        return {"GBps_median": round(rates[len(rates) // 2], 2), "GBps_best": round(rates[-1], 2)}

    def h2d():
        for k in range(streams):
            with torch.cuda.stream(st_in[k]):
                da[k].copy_(hs[k], non_blocking=True)

    def d2h():
        for k in range(streams):
            with torch.cuda.stream(st_out[k]):
                hd[k].copy_(db[k], non_blocking=True)

    def both():
        h2d()
        d2h()

    res["h2d"] = timed(h2d, n * streams)
    res["d2h"] = timed(d2h, n * streams)
    b = timed(both, 2 * n * streams)
    res["both_directions_at_once"] = {"GBps_sum_median": b["GBps_median"], "GBps_sum_best": b["GBps_best"],
                                      "GBps_per_direction_median": round(b["GBps_median"] / 2, 2)}
    # one stream, one copy at a time (what a single hipMemcpyAsync of a claim sees)
    one_in = timed(lambda: da[0].copy_(hs[0], non_blocking=True), n)
    one_out = timed(lambda: hd[0].copy_(db[0], non_blocking=True), n)
    res["h2d_one_stream"], res["d2h_one_stream"] = one_in, one_out
    del hs, hd, da, db
    torch.cuda.empty_cache()
    # device to device: 1 GiB, HBM traffic = 2 x the copied bytes
    g = 1 << 30
    x = torch.empty(g, dtype=torch.uint8, device=dev)
    y = torch.empty(g, dtype=torch.uint8, device=dev)
    x.random_(0, 256)
    d = timed(lambda: y.copy_(x), g)
    res["d2d_copy_1GiB"] = {"GBps_copied_median": d["GBps_median"], "GBps_copied_best": d["GBps_best"],
                            "GBps_hbm_traffic_median": round(2 * d["GBps_median"], 1), "GBps_hbm_traffic_best": round(2 * d["GBps_best"], 1),
                            "frac_of_8TBps_peak": round(2 * d["GBps_median"] / 8000.0, 3)}
    del x, y
    torch.cuda.empty_cache()
    links = pcie_links()
    res["pcie_links"] = links
    if links:
        l0 = links[min(device, len(links) - 1)]
        res["pcie_link_this_gpu"] = {"speed": l0.get("current_link_speed"), "width": l0.get("current_link_width"),
                                     "GBps_one_direction_raw": gbps_of_link(l0.get("current_link_speed"), l0.get("current_link_width"))}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=256)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = measure(a.mib, a.streams, a.reps)
    txt = json.dumps(r, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
