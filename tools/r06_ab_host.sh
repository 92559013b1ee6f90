#!/bin/bash
# round 6 A/B: kernel time with the results written into PINNED HOST memory (the product paths; PACKED entries) vs device memory, for variant libraries
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
for V in "$@"; do
  SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod_$V.so; [ "$V" = default ] && SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod.so
  for H in 1 0; do for rep in 1 2; do
    echo "== $V host_results=$H packed=$H (run $rep)"; KTIME_PACKED=$H KTIME_HOST_RESULTS=$H QZ_PLUGIN_SO=$SO timeout 300 python tools/ktime.py 1:131072:4096:system 3:131072:2048:system 2>&1 | grep "WG/CU"
  done; done
done > $O/ktime_host.txt
cat $O/ktime_host.txt | cut -c1-150
