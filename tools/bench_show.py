#!/usr/bin/env python3
"""prints the interesting keys of a bench.py JSON line (gpurun_out/*.json)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline", "cpu_baseline", "value_e2e", "unchanged_callers", "announced", "e2e_ceiling_replay",
          "product_multi_gpu", "frontend", "e2e_sweep_indicative", "kernel_other_levels", "level3", "level6", "level12_32k", "config5_shape_4MiB_frames_L3"):
    print(k, json.dumps(d.get(k))[:int(sys.argv[2]) if len(sys.argv) > 2 else 1200])
