#!/bin/bash
# A/B on the GPU box: kernel time at the BASELINE chain shapes for library variants (make variant NAME=...).
# usage: gpurun -- bash tools/ab_chain.sh "<name> <name> ..." ["shape shape ..."]      ("base" = lib/libqatseqprod.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
SHAPES=${2:-"6:131072:2048:system 12:32768:8192:weblog 12:131072:2048:system 9:131072:2048:system 5:131072:2048:system"}
for V in $1; do
  SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod_$V.so; [ $V = base ] && SO=$R/qat-zstd-plugin_amd/lib/libqatseqprod.so
  echo "=== $V"
  QZ_PLUGIN_SO=$SO timeout 900 python tools/ktime.py $SHAPES 2>&1 | grep -v amdgpu.ids
done
