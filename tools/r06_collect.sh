#!/bin/bash
# copy the round's judged evidence from gpurun_out/ (scratch) into profiles/ (tracked): usage tools/r06_collect.sh <final-call dir under gpurun_out>
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; F=gpurun_out/${1:-r06_final}
cp $F/bench.json profiles/r06_bench.json; cp $F/bench_details.json profiles/r06_bench_details.json
cp $F/gpu_suite.txt profiles/r06_gpu_suite.txt
c() { f=$(find gpurun_out/stats_$1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f profiles/$2; }
c l1 r06_kernel_stats.csv; c l6 r06_kernel_stats_level6_2048blocks.csv; c l12w r06_kernel_stats_level12_32k_weblog_8192blocks.csv
c l12 r06_kernel_stats_level12_2048blocks.csv; c l3 r06_kernel_stats_level3_4096blocks.csv; c l1rep r06_kernel_stats_level1_repcodes.csv
cp gpurun_out/pmc_l1/summary.txt profiles/r06_pmc_summary.txt
cp gpurun_out/pmc_l6/summary.txt profiles/r06_pmc_summary_level6_2048blocks.txt
cp gpurun_out/pmc_l12w/summary.txt profiles/r06_pmc_summary_level12_32k_weblog_8192blocks.txt
ls -la profiles/r06_*
