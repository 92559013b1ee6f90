#!/bin/bash
# Long fuzz run on a GPU box: tests/fuzz/fuzz_roundtrip.c against lib/libqatseqprod.so, six seeds x 400 iterations in the
# library's modes (default, opt-in look-ahead by process_vm_readv and by pipe, per-slot, repeat-aware).  usage: gpurun -- bash tools/fuzz_long.sh
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
gcc -O2 -g -std=c11 -D_POSIX_C_SOURCE=200809L -pthread -Iinclude -Ioracle -o /tmp/fuzz_gpu tests/fuzz/fuzz_roundtrip.c qat-zstd-plugin_amd/test/fuzzing/qatseqprodfuzzer.c -Lqat-zstd-plugin_amd/lib -lqatseqprod $Z -Wl,-rpath,$PWD/qat-zstd-plugin_amd/lib -Wl,-rpath,$(dirname $Z)
for seed in 101 102 103 104 105 106; do
  env=""; [ $seed = 103 ] && env="QZSTD_HIP_LOOKAHEAD=1"; [ $seed = 104 ] && env="QZSTD_HIP_COALESCE=0"; [ $seed = 105 ] && env="QZSTD_HIP_EXT_REPCODES=1"; [ $seed = 106 ] && env="QZSTD_HIP_LOOKAHEAD=2 QZSTD_HIP_TIMEOUT_MS=5000"
  env QZSTD_HIP_DEBUG=1 $env timeout 600 /tmp/fuzz_gpu $seed ${FUZZ_ITERS:-400} 2>&1 | grep -v "look-ahead is ON" | tail -6 || true
done
