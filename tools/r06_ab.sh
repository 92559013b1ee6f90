#!/bin/bash
# round 6 A/B on the GPU box: parity of the default build, then kernel times of the variant libraries named on the command line
# usage: tools/r06_ab.sh <tag> "<shapes for ktime.py>" variant...      (variant = the NAME of `make variant`; "default" = the product library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; TAG=$1; SHAPES=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
(QZ_PLUGIN_SO=${PARITY_SO:+$R/qat-zstd-plugin_amd/lib/libqatseqprod_$PARITY_SO.so} timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py ${EXTRA_TESTS:-} -x -q -m gpu 2>&1 | tail -60) > $O/parity.txt
for V in "$@"; do
  SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod_$V.so; [ "$V" = default ] && SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod.so
  for rep in 1 2; do
    echo "== $V (run $rep)"; QZ_PLUGIN_SO=$SO timeout 300 python tools/ktime.py $SHAPES 2>&1 | grep "WG/CU"
  done
done > $O/ktime.txt
(QZ_TIMING=1 QZ_BLOCKS=512 timeout 300 python tools/gpu_debug.py 2>&1 | tail -12) > $O/timing_512.txt
(QZ_TIMING=1 QZ_BLOCKS=256 timeout 300 python tools/gpu_debug.py 2>&1 | tail -12) > $O/timing_256.txt
cat $O/parity.txt | tail -3; cat $O/ktime.txt; cat $O/timing_512.txt
