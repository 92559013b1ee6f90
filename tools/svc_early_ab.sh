#!/bin/bash
# GPU box: unchanged callers (one ZSTD_compress2 per chunk, nothing announced) with and without progressive staging of service requests
# (QZSTD_HIP_SERVICE_EARLY=0: stage first, as in rounds 3-4), median pass of -P1 runs, P50 per call.   usage: tools/svc_early_ab.sh
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 32 << 20))
PY
BM=qat-zstd-plugin_amd/test/benchmark
for L in ${LEVELS:-1 6}; do for T in ${THREADS:-1 16}; do for rep in 1 2; do for E in 0 1; do
  echo -n "L$L T=$T early=$E: "; QZSTD_HIP_SERVICE_EARLY=$E $BM -m1 -t$T -l${LOOPS:-40} -c${CHUNK:-131072} -L$L -P1 /tmp/e2e.bin 2>&1 | grep -o "median [0-9.]* MB/s\|P50 [0-9.]*\|avg [0-9.]*\|Producer errors: [0-9]*" | tr '\n' ' '; echo
done; done; done; done
echo -n "software L1 T=1: "; $BM -m0 -t1 -l20 -c131072 -L1 -P1 /tmp/e2e.bin 2>&1 | grep -o "median [0-9.]* MB/s\|P50 [0-9.]*" | tr '\n' ' '; echo
