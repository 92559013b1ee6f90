python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/fe.bin","wb").write(K.system_corpus(512 << 20)[0])
PY
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
make -s -C qat-zstd-plugin_amd ZSTDLIB=$Z >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test frontbench replaybench ZSTDLIB=$Z >/dev/null 2>&1
for T in ${FE_THREADS:-13 14 15 16 17 18}; do echo -n "threads $T: "; qat-zstd-plugin_amd/test/frontbench -t$T -l12 -c131072 -L1 -s2 -m1 /tmp/fe.bin | grep -o "wall-clock [0-9.]* MB/s\|median [0-9.]* min [0-9.]* max [0-9.]*" | tr '\n' ' '; echo; done
for S in ${FE_SEGS:-1 4 8}; do echo -n "threads 16 seg $S MiB: "; qat-zstd-plugin_amd/test/frontbench -t16 -l12 -c131072 -L1 -s$S -m1 /tmp/fe.bin | grep -o "median [0-9.]* min [0-9.]* max [0-9.]*" | tr '\n' ' '; echo; done
