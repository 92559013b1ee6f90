#!/bin/bash
# GPU box: where a worker of the batch front-end spends its time (per state: staging, queueing, copy / launch calls, waits), A/B of knobs
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
d = K.system_corpus(512 << 20)[0]
open("/tmp/fe.bin","wb").write(d); open("/tmp/fe2.bin","wb").write(d + d)
PY
make -s -C qat-zstd-plugin_amd >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test frontbench >/dev/null 2>&1
FB=qat-zstd-plugin_amd/test/frontbench
run() { echo "== seg ${SEG:-4} ${BUF:-} $*"; env "$@" QZSTD_HIP_DEBUG=2 $FB -t${T:-16} -l10 -c131072 -L1 -s${SEG:-4} -m1 ${BUF:-/tmp/fe.bin} 2>&1 | grep -o "median [0-9.]* min [0-9.]* max [0-9.]*\|[0-9]* hint(s): .*\|PASS\|FAIL" | sort | uniq -c | sort -rn | head -${ROWS:-4}; }
run A=0
run QZSTD_HIP_HINT_DIRECT=1
ROWS=3
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
SEG=2 run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
SEG=1 run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
ROWS=2
SEG=8 run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
T=18 run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
BUF=/tmp/fe2.bin SEG=2 run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
BUF=/tmp/fe2.bin SEG=4 run GPU_MAX_HW_QUEUES=16 QZSTD_HIP_HINT_DIRECT=1
BUF=/tmp/fe2.bin SEG=4 run GPU_MAX_HW_QUEUES=16
