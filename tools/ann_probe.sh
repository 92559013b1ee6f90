python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/s.bin","wb").write(K.system_corpus(32 << 20)[0])
PY
make -s -C qat-zstd-plugin_amd >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test benchmark >/dev/null 2>&1
BM=qat-zstd-plugin_amd/test/benchmark
for T in 16 20 32; do for H in 2 4; do echo -n "announced -H$H threads $T: "; $BM -m1 -t$T -l30 -c131072 -L1 -H$H -P1 /tmp/s.bin 2>&1 | grep -o "median [0-9.]* MB/s, min [0-9.]*, max [0-9.]*\|P50 [0-9.]* *P75 [0-9.]* *P99 [0-9.]*" | tr '\n' ' '; echo; done; done
