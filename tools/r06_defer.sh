#!/bin/bash
# round 6, the deferred repeat-aware parse: parity of the default build on the GPU box, then kernel times of the libraries named on the command line
# usage: tools/r06_defer.sh <tag> "<shapes for ktime.py>" variant...      (variant = the NAME of `make variant`; "default" = the product library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; TAG=$1; SHAPES=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py ${EXTRA_TESTS:-} -x -q -m gpu 2>&1 | tail -40) > $O/parity.txt
for V in "$@"; do
  SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod_$V.so; [ "$V" = default ] && SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod.so
  echo "== $V"; QZ_PLUGIN_SO=$SO timeout 600 python tools/ktime.py $SHAPES 2>&1 | grep -E "WG/CU|Error|error|assert"
done > $O/ktime.txt
tail -5 $O/parity.txt; cat $O/ktime.txt
