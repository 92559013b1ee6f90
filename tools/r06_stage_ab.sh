#!/bin/bash
# round 6 A/B: the announcements' staging copy with streaming stores (QZSTD_HIP_STAGE_NT) — the batch front-end and the announcing benchmark tool
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out/${1:-r06_stage}; mkdir -p $O
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
d = K.system_corpus(512 << 20)[0]
open("/tmp/fb.bin","wb").write(d); open("/tmp/fs.bin","wb").write(d[:48 << 20])
PY
Z=$(python tools/qz_bind.py --libzstd)
make -C qat-zstd-plugin_amd/test frontbench benchmark ZSTDLIB=$Z > /dev/null 2>&1
for rep in 1 2 3; do for NT in 0 1; do
  echo "== QZSTD_HIP_STAGE_NT=$NT front-end 17 threads"; QZSTD_HIP_STAGE_NT=$NT qat-zstd-plugin_amd/test/frontbench -t17 -l24 -c131072 -L1 -s2 -m1 /tmp/fb.bin 2>&1 | grep -E "passes MB"
done; done > $O/frontbench.txt
for rep in 1 2; do for NT in 0 1; do
  echo "== QZSTD_HIP_STAGE_NT=$NT benchmark -m1 -H2 (verified announcements), 16 threads"; QZSTD_HIP_STAGE_NT=$NT qat-zstd-plugin_amd/test/benchmark -m1 -t16 -l30 -c131072 -L1 -H2 -P1 /tmp/fs.bin 2>&1 | grep -E "asses|aggregate" | cut -c1-200
done; done > $O/benchmark_h2.txt
cat $O/frontbench.txt $O/benchmark_h2.txt
