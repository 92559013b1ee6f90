python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 32 << 20))
PY
BM=qat-zstd-plugin_amd/test/benchmark
for T in 16 32; do for rep in 1 2; do for I in 4096 8192; do
  echo -n "T=$T item=$I: "; QZSTD_HIP_SERVICE_ITEM=$I $BM -m1 -t$T -l40 -c131072 -L1 -P1 /tmp/e2e.bin 2>&1 | grep -o "median [0-9.]* MB/s\|P50 [0-9.]*" | tr '\n' ' '; echo
done; done; done
