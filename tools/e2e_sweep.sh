#!/bin/bash
# End-to-end sweep of the C benchmark tool on the GPU box: software vs plugin (unchanged callers, announcements,
# opt-in transparent look-ahead) over thread counts.  Usage: tools/e2e_sweep.sh [MiB per thread, default 16] [level, default 1] [chunk]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MB=${1:-16}; L=${2:-1}; C=${3:-131072}
cd $R
ZL=$(python - <<'PY'
import sys; sys.path.insert(0, "tools"); import qz_bind as B; print(B.find_libzstd())
PY
)
make -C qat-zstd-plugin_amd/test benchmark ZSTDLIB=$ZL >/dev/null 2>&1
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", $MB << 20))
PY
BM=qat-zstd-plugin_amd/test/benchmark
echo "nproc $(nproc) level $L chunk $C, $MB MiB per thread"
for T in ${THREADS:-1 16 64 128 256}; do
  for cfg in "sw:-m0" "plain:-m1" "hint:-m1 -H4"; do
    name=${cfg%%:*}; args=${cfg#*:}
    out=$(env $env timeout 300 $BM $args -t$T -l${LOOPS:-2} -c$C -L$L /tmp/e2e.bin 2>&1 | grep "aggregate compression")
    echo "T=$T $name: $(echo $out | sed 's/.*aggregate compression //; s/decompression.*//')"
  done
done
