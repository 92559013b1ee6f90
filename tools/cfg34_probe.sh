#!/bin/bash
# BASELINE config 3 / 4 shapes end to end on a GPU box (C benchmark tool): level 6 on 32 MiB of Zipf text in 128 KiB
# chunks, level 12 on 32 MiB of web-log lines in 32 KiB chunks; software zstd beside the plugin.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python - <<'PY'
import sys; sys.path.insert(0,'tools')
import qz_corpus as K
open('/tmp/weblog.bin','wb').write(K.weblog(4, 32<<20))
open('/tmp/enw.bin','wb').write(K.text(3, 32<<20))
PY
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
make -C qat-zstd-plugin_amd/test benchmark ZSTDLIB=$Z >/dev/null
cd qat-zstd-plugin_amd/test
echo "== cfg4 sw L12 32K"; ./benchmark -m0 -t16 -l1 -c32K -L12 /tmp/weblog.bin 2>&1 | grep -E "aggregate|Thread 0"
echo "== cfg4 plugin L12 32K H16"; ./benchmark -m1 -t16 -l3 -c32K -L12 -H16 /tmp/weblog.bin 2>&1 | grep -E "aggregate|Thread 0"
echo "== cfg3 sw L6 128K text"; ./benchmark -m0 -t16 -l1 -c128K -L6 /tmp/enw.bin 2>&1 | grep -E "aggregate|Thread 0"
echo "== cfg3 plugin L6 E1 REP"; QZSTD_HIP_EXT_REPCODES=1 ./benchmark -m1 -t16 -l3 -c128K -L6 -E1 -H16 /tmp/enw.bin 2>&1 | grep -E "aggregate|Thread 0"
echo "== cfg3 plugin L6 default"; ./benchmark -m1 -t16 -l3 -c128K -L6 -H16 /tmp/enw.bin 2>&1 | grep -E "aggregate|Thread 0"
