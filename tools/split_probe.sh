#!/bin/bash
# Per-block path with and without segment work items (QZSTD_HIP_SPLIT_BLOCKS) vs software, levels 1 and 3.  usage: gpurun -- bash tools/split_probe.sh
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 32 << 20))
PY
BM=qat-zstd-plugin_amd/test/benchmark
for L in ${LEVELS:-1 3}; do
echo -n "L$L sw T=16: "; $BM -m0 -t16 -l4 -c131072 -L$L /tmp/e2e.bin 2>&1 | grep -o "[0-9.]* MB/s by the wall"
for T in 1 16 32 64; do for sp in 0 1; do echo -n "L$L T=$T split=$sp: "; QZSTD_HIP_SPLIT_BLOCKS=$sp $BM -m1 -t$T -l4 -c131072 -L$L /tmp/e2e.bin 2>&1 | grep -o "[0-9.]* MB/s by the wall\|P50 [0-9.]*" | tr '\n' ' '; echo; done; done; done
