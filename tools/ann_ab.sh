python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 32 << 20))
PY
mkdir -p /tmp/bl; cp ${BASE_SO:-qat-zstd-plugin_amd/lib/libqatseqprod_base.so} /tmp/bl/libqatseqprod.so  # a library built from the revision to compare with (make variant NAME=base in a checkout of it)
BM=qat-zstd-plugin_amd/test/benchmark
for rep in 1 2 3; do
  echo -n "new  : "; $BM -m1 -H2 -t16 -l60 -c131072 -L1 -P1 /tmp/e2e.bin 2>&1 | grep -o "median [0-9.]* MB/s\|P50 [0-9.]*\|Producer errors: [0-9]*" | tr '\n' ' '; echo
  echo -n "base : "; LD_LIBRARY_PATH=/tmp/bl $BM -m1 -H2 -t16 -l60 -c131072 -L1 -P1 /tmp/e2e.bin 2>&1 | grep -o "median [0-9.]* MB/s\|P50 [0-9.]*\|Producer errors: [0-9]*" | tr '\n' ' '; echo
done
