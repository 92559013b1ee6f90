#!/bin/bash
# A/B on the GPU box: parity + kernel ms per level for library variants.  usage: gpurun -- bash tools/ring_ab.sh "<so> <so> ..." "<levels>"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for SO in $1; do
  echo "=== $SO"
  QZ_PLUGIN_SO=$R/qat-zstd-plugin_amd/lib/$SO timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -1
  for LV in ${2:-1 3 6}; do
    B=8192; case $LV in 3|4) B=4096;; 5|6|7|8|9|10|11|12|0x101) B=2048;; esac
    echo -n "level $LV ($B blocks): "
    QZ_PLUGIN_SO=$R/qat-zstd-plugin_amd/lib/$SO timeout 300 python bench.py --kernel-only --steps 6 --warmup 2 --level $LV --blocks $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernel_ms_avg'], 'ms', d.get('error_blocks'))"
  done
done
