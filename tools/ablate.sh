#!/bin/bash
# Kernel time of the default bench workload under the profiling build's ablation switches (GPU box).
# usage: tools/ablate.sh "<level>" "<bits> <bits> ..."   (needs `make debug`; QZSTD_HIP_ABLATE bits: csrc/qzstd_kernels.hip)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LV=${1:-1}; shift
for A in ${1:-0}; do
  echo -n "level $LV ablate $A: "
  QZ_PLUGIN_SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod_dbg.so QZSTD_HIP_ABLATE=$A timeout 300 python $R/bench.py --kernel-only --steps 6 --warmup 2 --level $LV ${BLOCKS:+--blocks $BLOCKS} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernel_ms_avg'], 'ms', d.get('error_blocks'))"
done
