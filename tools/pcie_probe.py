import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("qz_bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import qz_bind as B
plug = B.Plugin()
data, _ = bench.load_corpus("system", 131072 * 8192)
for mode in ("memcpy", "kernel", "memcpy", "kernel"):
    os.environ["QZ_BENCH_PCIE_COPY"] = mode
    for cb in (512, 1024):
        r = bench.pcie_pipeline_leg(plug, data, 131072, 1, 0, chunk_blocks=cb)
        print(mode, cb, r.get("GBps_input_per_gpu"), r.get("GBps_best_pass"), r.get("error"))
