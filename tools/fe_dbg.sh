#!/bin/bash
# GPU box: where a worker of the batch front-end spends its time (per state: staging, queueing, copy / launch calls, waits), A/B of knobs
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
d = K.system_corpus(512 << 20)[0]
open("/tmp/fe.bin","wb").write(d); open("/tmp/fe2.bin","wb").write(d + d)
PY
make -s -C qat-zstd-plugin_amd >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test frontbench >/dev/null 2>&1
FB=qat-zstd-plugin_amd/test/frontbench
run() { echo "== seg ${SEG:-4} threads ${T:-16} L${LV:-1} ${BUF:-} $*"; env "$@" QZSTD_HIP_DEBUG=2 $FB -t${T:-16} -l${LOOPS:-10} -c131072 -L${LV:-1} -s${SEG:-4} -m1 ${BUF:-/tmp/fe.bin} 2>&1 | grep -o "median [0-9.]* min [0-9.]* max [0-9.]*\|[0-9]* hint(s), .*\|PASS\|FAIL\|[0-9]* block(s) from announcements, [0-9]* per block\|roducer errors: [0-9]*" | sort | uniq -c | sort -rn | head -${ROWS:-5}; }
if [ $# -gt 0 ]; then for CFG in "$@"; do eval "$CFG"; done; exit 0; fi
SEG=2 run QZSTD_FRONT_AHEAD=2
SEG=2 run QZSTD_FRONT_AHEAD=2 QZSTD_HIP_HINT_DIRECT=0
SEG=2 run QZSTD_FRONT_AHEAD=2 QZSTD_HIP_HINT_FLAGS=0
SEG=2 run QZSTD_FRONT_AHEAD=2 QZSTD_HIP_HINT_FLAGS=0 QZSTD_HIP_HINT_DIRECT=0
ROWS=3
SEG=4 run QZSTD_FRONT_AHEAD=1
SEG=4 run QZSTD_FRONT_AHEAD=2
SEG=4 run QZSTD_FRONT_AHEAD=2 QZSTD_HIP_HINT_DIRECT=0
SEG=1 run QZSTD_FRONT_AHEAD=3
SEG=1 run QZSTD_FRONT_AHEAD=3 QZSTD_HIP_HINT_DIRECT=0
SEG=2 T=18 run QZSTD_FRONT_AHEAD=2
BUF=/tmp/fe2.bin SEG=2 run QZSTD_FRONT_AHEAD=2
LV=3 SEG=2 run QZSTD_FRONT_AHEAD=2
LV=3 SEG=2 run QZSTD_FRONT_AHEAD=2 QZSTD_HIP_HINT_DIRECT=0
LV=6 SEG=2 LOOPS=3 run QZSTD_FRONT_AHEAD=2
