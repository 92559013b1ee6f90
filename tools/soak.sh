#!/bin/bash
# Soak test on a GPU box: the C benchmark tool in many configurations, every run under a timeout; prints one line per
# run and a final verdict.  usage: tools/soak.sh [MiB of corpus, default 24]
MB=${1:-24}
cd "$(dirname "$0")/.."
python - "$MB" <<'PY'
import sys; sys.path.insert(0,'tools')
import qz_corpus as K
n=int(sys.argv[1])<<20
open('/tmp/soak_sys.bin','wb').write(K.by_name('system', n))
open('/tmp/soak_mix.bin','wb').write(K.by_name('mix', n//2, seed=9))
PY
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
make -C qat-zstd-plugin_amd/test benchmark ZSTDLIB=$Z >/dev/null
cd qat-zstd-plugin_amd/test
fail=0; runs=0
run() { # env... -- args
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local out; out=$(env "${envs[@]}" timeout 120 ./benchmark "$@" 2>&1); local rc=$?
  local pass; pass=$(echo "$out" | grep -c PASS)
  runs=$((runs+1))
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "FAIL rc=$rc: ${envs[*]} $*"; echo "$out" | tail -3; else echo "ok   pass=$pass: ${envs[*]} $*"; fi
}
for round in 1 2; do
 for L in 1 2 3 5 6 9 12; do
  for T in 1 7 16 40; do
   run A=1 -- -m1 -t$T -l2 -c128K -L$L /tmp/soak_sys.bin
  done
  run A=1 -- -m1 -t12 -l2 -c64K -L$L -H2 /tmp/soak_mix.bin
  run QZSTD_HIP_EXT_REPCODES=1 -- -m1 -t9 -l2 -c128K -L$L -E1 -H8 /tmp/soak_sys.bin
  run QZSTD_HIP_SERVICE=0 -- -m1 -t20 -l1 -c32K -L$L /tmp/soak_mix.bin
  run QZSTD_HIP_TIMEOUT_MS=1 -- -m1 -t6 -l1 -c128K -L$L -F1 /tmp/soak_mix.bin
  run QZSTD_HIP_COALESCE=0 QZSTD_HIP_SLOTS=6 -- -m1 -t10 -l1 -c100000 -L$L /tmp/soak_mix.bin
 done
done
echo "soak: $runs runs, $fail failed"
