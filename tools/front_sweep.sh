#!/bin/bash
# Front-end vs the replay ceiling on the GPU box: worker threads x segment size.  usage: tools/front_sweep.sh [MiB of corpus, default 256]
MB=${1:-256}
cd "$(dirname "$0")/.."
python - "$MB" <<'PY'
import sys; sys.path.insert(0,'tools')
import qz_corpus as K
open('/tmp/fs_sys.bin','wb').write(K.by_name('system', int(sys.argv[1])<<20))
PY
Z=$(python -c "import sys; sys.path.insert(0,'tools'); import qz_bind as B; print(B.find_libzstd())")
make -C qat-zstd-plugin_amd ZSTDLIB=$Z >/dev/null 2>&1
make -C qat-zstd-plugin_amd/test frontbench replaybench ZSTDLIB=$Z >/dev/null 2>&1
cd qat-zstd-plugin_amd/test
for T in 16 20 24 32; do
  echo -n "T=$T ceiling: "; ./replaybench -t$T -l3 -c128K -L1 /tmp/fs_sys.bin | grep -o "[0-9.]* MB/s wall (best pass [0-9.]*)"
  for S in 1 2 4 8; do
    echo -n "T=$T seg=${S}MiB front-end: "; ./frontbench -t$T -l3 -c128K -L1 -s$S -m1 /tmp/fs_sys.bin | grep -o "wall-clock [0-9.]* MB/s (mean of [0-9]* passes; best [0-9.]* MB/s)"
  done
done
