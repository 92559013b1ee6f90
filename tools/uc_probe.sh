#!/bin/bash
# GPU box: unchanged callers (nothing announced) at level 1 — threads beyond the cores x how long a waiting caller polls before it naps
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/s.bin","wb").write(K.system_corpus(32 << 20)[0])
PY
make -s -C qat-zstd-plugin_amd >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test benchmark >/dev/null 2>&1
BM=qat-zstd-plugin_amd/test/benchmark
for SPIN in ${SPINS:-default 400 10}; do for T in ${THREADS:-16 24 32 48}; do echo -n "level ${LV:-1} spin $SPIN us threads $T: "; env $([ $SPIN = default ] && echo A=1 || echo QZSTD_HIP_SERVICE_SPIN_US=$SPIN) $BM -m1 -t$T -l${LOOPS:-30} -c${CH:-131072} -L${LV:-1} -P1 /tmp/s.bin 2>&1 | grep -o "median [0-9.]* MB/s, min [0-9.]*, max [0-9.]*\|P50 [0-9.]* *P75 [0-9.]* *P99 [0-9.]*" | tr '\n' ' '; echo; done; done
