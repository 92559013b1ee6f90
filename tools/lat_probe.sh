#!/bin/bash
# Unchanged callers (plain) vs announced (-H2) vs software at 1 and N threads: MB/s by the wall clock + P50 latency per call.
# usage: gpurun -- bash tools/lat_probe.sh   (env: LEVELS="1 3", THREADS="1 16", EXTRA="VAR=val ...")
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 32 << 20))
PY
BM=qat-zstd-plugin_amd/test/benchmark
make -s -C qat-zstd-plugin_amd/test benchmark >/dev/null 2>&1
for L in ${LEVELS:-1}; do
for T in ${THREADS:-1 16}; do
  echo -n "L$L T=$T software : "; $BM -m0 -t$T -l${LOOPS:-4} -c${CHUNK:-131072} -L$L /tmp/e2e.bin 2>&1 | grep -o "[0-9.]* MB/s by the wall\|P50 [0-9.]*" | tr '\n' ' '; echo
  echo -n "L$L T=$T announced: "; env $EXTRA $BM -m1 -H2 -t$T -l${LOOPS:-4} -c${CHUNK:-131072} -L$L /tmp/e2e.bin 2>&1 | grep -o "[0-9.]* MB/s by the wall\|P50 [0-9.]*" | tr '\n' ' '; echo
  echo -n "L$L T=$T plain    : "; env $EXTRA $BM -m1 -t$T -l${LOOPS:-4} -c${CHUNK:-131072} -L$L /tmp/e2e.bin 2>&1 | grep -o "[0-9.]* MB/s by the wall\|P50 [0-9.]*" | tr '\n' ' '; echo
done; done
