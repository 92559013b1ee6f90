import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("qz_bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import qz_bind as B
plug = B.Plugin()
data, _ = bench.load_corpus("system", 131072 * 8192)
# usage: python tools/pcie_probe.py [level ...]   the PCIe-inclusive pipeline of the C ABI, 16-byte against PACKED result entries, copy engine against copy kernel
for level in [int(x, 0) for x in sys.argv[1:]] or [1]:
    for packed in ("0", "1", "0", "1"):
        os.environ["QZ_BENCH_PCIE_PACKED"] = packed
        for mode in ("memcpy", "kernel"):
            os.environ["QZ_BENCH_PCIE_COPY"] = mode
            for cb, depth in ((512, 3), (256, 4)):
                r = bench.pcie_pipeline_leg(plug, data, 131072, level, 0, chunk_blocks=cb, depth=depth)
                print("level %#x entries %2d B  %-6s chunks of %4d blocks x %d in flight: %s GB/s of input (best pass %s), %s MB of results per pass %s"
                      % (level, 8 if packed == "1" else 16, mode, cb, depth, r.get("GBps_input_per_gpu"), r.get("GBps_best_pass"),
                         (r.get("result_bytes_per_pass") or 0) >> 20, r.get("error") or ""), flush=True)
