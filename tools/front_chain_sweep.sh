#!/bin/bash
# GPU box: the batch front-end at the chain levels — worker threads x claim size x claims announced ahead (median pass).
# usage: tools/front_chain_sweep.sh     (env: LEVELS="6 12", 512 MiB system corpus at 128 KiB chunks; level 12: 256 MiB web-log at 32 KiB)
cd "$(dirname "$0")/.."
python - <<'PY'
import sys; sys.path.insert(0,'tools')
import qz_corpus as K
open('/tmp/fc_sys.bin','wb').write(K.system_corpus(512 << 20)[0])
open('/tmp/fc_web.bin','wb').write(K.weblog(4, 64 << 20) * 4)
PY
FB=qat-zstd-plugin_amd/test/frontbench
for L in ${LEVELS:-6 12}; do
  if [ $L = 12 ]; then F=/tmp/fc_web.bin; C=32768; else F=/tmp/fc_sys.bin; C=131072; fi
  for T in ${THREADS:-17 24 32}; do for S in ${SEGS:-2 4 8}; do for A in ${AHEADS:-3}; do
    echo -n "L$L T=$T seg=${S}MiB ahead=$A: "; QZSTD_FRONT_AHEAD=$A $FB -t$T -l6 -c$C -L$L -s$S -m1 $F | grep -o "passes MB/s: median [0-9.]* min [0-9.]* max [0-9.]*\|producer errors: [0-9]*" | tr '\n' ' '; echo
  done; done; done
done
