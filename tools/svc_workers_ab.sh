python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 32 << 20))
PY
BM=qat-zstd-plugin_amd/test/benchmark
for T in 16 32; do for cfg in "A=1" "QZSTD_HIP_SERVICE_MULTI=0" "QZSTD_HIP_SERVICE_MULTI=0 QZSTD_HIP_SERVICE_WORKERS=512" "QZSTD_HIP_SERVICE_MULTI=0 QZSTD_HIP_SERVICE_WORKERS=512 QZSTD_HIP_SERVICE_ITEM=8192"; do for rep in 1 2; do
  echo -n "T=$T [$cfg]: "; env $cfg $BM -m1 -t$T -l40 -c131072 -L1 -P1 /tmp/e2e.bin 2>&1 | grep -o "median [0-9.]* MB/s\|P50 [0-9.]*\|avg [0-9.]*\|Producer errors: [0-9]*" | tr '\n' ' '; echo
done; done; done
