#!/bin/bash
# End-to-end sweep of the C benchmark tool on a GPU box: software zstd vs the plugin, with and
# without look-ahead hints.  Usage: tools/e2e_sweep.sh [MiB of corpus, default 64] [level, default 1]
MB=${1:-64}; LV=${2:-1}
cd "$(dirname "$0")/.."
python - "$MB" <<'PY'
import sys; sys.path.insert(0,'tools')
import qz_corpus as K
open('/tmp/corpus.bin','wb').write(K.by_name('system', int(sys.argv[1])<<20))
PY
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
make -C qat-zstd-plugin_amd/test benchmark ZSTDLIB=$Z >/dev/null
cd qat-zstd-plugin_amd/test
echo "cores: $(nproc)"
for T in 1 16 32; do
  echo "== software zstd  t$T"; ./benchmark -m0 -t$T -l2 -c128K -L$LV /tmp/corpus.bin 2>&1 | tail -2
done
for H in 0 1 2 8 16; do
  for T in 1 16 32; do
    echo "== plugin H$H t$T"; ./benchmark -m1 -t$T -l2 -c128K -L$LV -H$H /tmp/corpus.bin 2>&1 | tail -2
  done
done
