#!/bin/bash
# round 6: confirmation of HEAD in one GPU call — the whole -m gpu suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out/${1:-r06_confirm}; mkdir -p $O
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/gpu_suite.txt
(timeout 600 python -c 'import __graft_entry__ as g; g.smoke(); print("SMOKE OK")' 2>&1 | tail -5) > $O/smoke.txt
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cp bench_details.json $O/bench_details.json)
tail -3 $O/gpu_suite.txt; tail -2 $O/smoke.txt; tail -c 1500 $O/bench.json
