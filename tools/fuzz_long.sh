#!/bin/bash
# Long fuzz run on a GPU box: tests/fuzz/fuzz_roundtrip.c against lib/libqatseqprod.so, six seeds x 400 iterations in the
# library's modes (default, batches only, per-slot, repeat-aware, long time-out); every frame is also
# compared with the frame libzstd builds from the oracle's sequences.  A seed that fails prints FAILED and the script exits
# non-zero.  usage: gpurun -- bash tools/fuzz_long.sh
set -e -o pipefail
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
gcc -O2 -g -std=c11 -D_POSIX_C_SOURCE=200809L -pthread -Iinclude -Ioracle -o /tmp/fuzz_gpu tests/fuzz/fuzz_roundtrip.c qat-zstd-plugin_amd/test/fuzzing/qatseqprodfuzzer.c oracle/qzstd_oracle.c -Lqat-zstd-plugin_amd/lib -lqatseqprod $Z -Wl,-rpath,$PWD/qat-zstd-plugin_amd/lib -Wl,-rpath,$(dirname $Z)
failed=0
for seed in ${FUZZ_SEEDS:-101 102 103 104 105 106}; do
  env=""; [ $seed = 103 ] && env="QZSTD_HIP_SERVICE=0"; [ $seed = 104 ] && env="QZSTD_HIP_COALESCE=0"; [ $seed = 105 ] && env="QZSTD_HIP_EXT_REPCODES=1"; [ $seed = 106 ] && env="QZSTD_HIP_TIMEOUT_MS=5000"
  set +e
  env QZSTD_HIP_DEBUG=1 $env timeout ${FUZZ_TIMEOUT:-900} /tmp/fuzz_gpu $seed ${FUZZ_ITERS:-400} 3072 ${FUZZ_ORACLE_EVERY:-1} > /tmp/fuzz_$seed.log 2>&1
  rc=$?
  set -e
  tail -6 /tmp/fuzz_$seed.log || true
  if [ $rc -ne 0 ]; then echo "FAILED seed $seed (rc $rc, env: $env)"; failed=$((failed+1)); fi
done
echo "fuzz_long: $failed seed(s) failed"
[ $failed -eq 0 ]
