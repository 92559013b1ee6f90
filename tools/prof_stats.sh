#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench workload (run on the GPU box via gpurun).
# usage: tools/prof_stats.sh <tag>     -> gpurun_out/stats_<tag>/  (copy the summary into profiles/)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $R/bench.py --steps 10 --warmup 2 --kernel-only "$@" > $OUT/run.log 2>&1
find $OUT -name '*kernel_stats.csv' -exec cat {} \;
