#!/bin/bash
# round 6, first GPU call: the new tests, the timing picture of the level-1 kernel (profiling build), the box's ceilings, the bench line, the full suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out/r06_call1; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_replicated_devices.py tests/test_gpu_hidden_device.py -x -q -m gpu 2>&1 | tail -30) > $O/new_tests.txt
(QZ_TIMING=1 QZ_HWID=1 QZ_BLOCKS=512 timeout 300 python tools/gpu_debug.py 2>&1 | tail -30) > $O/timing_level1_512blocks.txt
(QZ_TIMING=1 QZ_BLOCKS=256 timeout 300 python tools/gpu_debug.py 2>&1 | tail -30) > $O/timing_level1_256blocks.txt
(timeout 300 python tools/box_ceilings.py --out $O/box_ceilings.json 2>&1 | tail -5) > $O/box_ceilings.log
(timeout 300 python tools/ktime.py 1:131072:8192:system 2:131072:4096:system 3:131072:4096:system 2>&1 | tail -5) > $O/ktime.txt
(timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cp bench_details.json $O/bench_details.json)
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/gpu_suite.txt
