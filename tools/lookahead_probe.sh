#!/bin/bash
# GPU box: the opt-in transparent look-ahead (QZSTD_HIP_LOOKAHEAD) against the default per-block path, unchanged callers
# usage: LV=12 CH=32768 tools/lookahead_probe.sh
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/s.bin","wb").write(K.system_corpus(32 << 20)[0])
PY
make -s -C qat-zstd-plugin_amd >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test benchmark >/dev/null 2>&1
BM=qat-zstd-plugin_amd/test/benchmark
for E in "A=1" "QZSTD_HIP_LOOKAHEAD=1" "QZSTD_HIP_LOOKAHEAD=2"; do echo -n "level ${LV:-1} chunk ${CH:-131072} $E: "; env $E $BM -m1 -t${T:-16} -l${LOOPS:-40} -c${CH:-131072} -L${LV:-1} -P1 /tmp/s.bin 2>&1 | grep -o "median [0-9.]* MB/s, min [0-9.]*, max [0-9.]*\|P50 [0-9.]* *P75 [0-9.]* *P99 [0-9.]*" | tr '\n' ' '; echo; done
