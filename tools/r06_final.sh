#!/bin/bash
# round 6: the evidence of the final tree in one GPU call — the whole -m gpu suite, the profiling round (rocprofv3 stats + PMC passes), the bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out/${1:-r06_final}; mkdir -p $O
(timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/gpu_suite.txt
(timeout 1800 bash tools/prof_round.sh 2>&1) > $O/prof_round.txt
(timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cp bench_details.json $O/bench_details.json)
tail -3 $O/gpu_suite.txt; tail -c 1500 $O/bench.json
