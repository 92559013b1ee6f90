#!/bin/bash
# GPU box: the batch front-end (value's leg) and the replay ceiling with the two libzstd 1.5.7 builds of the image — the normal build inside
# libarrow.so through tools/zstdshim (default) and Pillow's exported copy (QZ_ZSTD_NO_SHIM=1) — over worker threads and segment sizes.
# usage: tools/fe_fast.sh   (env FE_THREADS="12 16 18 20 24", FE_SEGS="1 2 4")
set -o pipefail
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/fe.bin","wb").write(K.system_corpus(512 << 20)[0])
PY
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for NOSHIM in 0 1; do
  export QZ_ZSTD_NO_SHIM=$NOSHIM
  Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
  echo "== libzstd: $Z"
  make -s -C qat-zstd-plugin_amd ZSTDLIB=$Z >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test frontbench replaybench benchmark ZSTDLIB=$Z >/dev/null 2>&1
  head -c $((256 << 20)) /tmp/fe.bin > /tmp/fq.bin
  echo -n "replay ceiling 16 threads: "; qat-zstd-plugin_amd/test/replaybench -t16 -l8 -c131072 -L1 /tmp/fq.bin | grep -o "[0-9.]* MB/s wall (best pass [0-9.]*)"
  echo -n "software level 1, 16 threads: "; head -c $((32 << 20)) /tmp/fe.bin > /tmp/fs.bin; qat-zstd-plugin_amd/test/benchmark -m0 -t16 -l8 -c131072 -L1 -P1 /tmp/fs.bin 2>&1 | grep -o "median [0-9.]* MB/s, min [0-9.]*, max [0-9.]*\|[0-9.]* MB/s by the wall clock" | tr '\n' ' '; echo
  TH="${FE_THREADS:-12 14 16 18 20 24}"; [ $NOSHIM = 1 ] && TH="16 18"
  for T in $TH; do echo -n "threads $T seg 2: "; qat-zstd-plugin_amd/test/frontbench -t$T -l12 -c131072 -L1 -s2 -m1 /tmp/fe.bin | grep -o "wall-clock [0-9.]* MB/s\|median [0-9.]* min [0-9.]* max [0-9.]*\|PASS\|FAIL" | tr '\n' ' '; echo; done
  [ $NOSHIM = 1 ] && continue
  for S in ${FE_SEGS:-1 4 8}; do for T in 16 18; do echo -n "threads $T seg $S MiB: "; qat-zstd-plugin_amd/test/frontbench -t$T -l12 -c131072 -L1 -s$S -m1 /tmp/fe.bin | grep -o "median [0-9.]* min [0-9.]* max [0-9.]*" | tr '\n' ' '; echo; done; done
done
