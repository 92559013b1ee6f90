#!/bin/bash
# round 6: PACKED result entries (qzstd_hip.h: QZSTD_HIP_MARK_COMPACT) — parity, then what they buy where results cross PCIe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out/${1:-r06_packed}; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_plugin.py tests/test_gpu_frontend.py tests/test_gpu_replicated_devices.py -x -q -m gpu 2>&1 | tail -25) > $O/tests.txt
for P in 0 1; do for H in 1 0; do
  echo "== packed=$P host_results=$H"; KTIME_PACKED=$P KTIME_HOST_RESULTS=$H timeout 300 python tools/ktime.py 1:131072:4096:system 3:131072:2048:system 6:131072:1024:system 12:32768:4096:weblog 2>&1 | grep "WG/CU"
done; done > $O/ktime.txt
(timeout 600 python tools/pcie_probe.py 1 2>&1 | grep "level") > $O/pcie.txt
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/fb.bin","wb").write(K.system_corpus(512 << 20)[0])
PY
make -C qat-zstd-plugin_amd/test frontbench ZSTDLIB=$(python tools/qz_bind.py --libzstd) > /dev/null 2>&1
for C in 0 1 0 1; do
  echo "== QZSTD_HIP_HINT_COMPACT=$C"; QZSTD_HIP_HINT_COMPACT=$C qat-zstd-plugin_amd/test/frontbench -t17 -l20 -c131072 -L1 -s2 -m1 /tmp/fb.bin 2>&1 | grep -E "wall-clock|passes MB|errors"
done > $O/frontbench.txt
for C in 0 1; do echo "== QZSTD_HIP_HINT_COMPACT=$C"; QZSTD_HIP_HINT_COMPACT=$C THREADS="8 16" SEGS="2 4" bash tools/hostpath_gpu.sh 2>&1 | tail -4; done > $O/hostpath.txt
tail -4 $O/tests.txt; cat $O/ktime.txt | cut -c1-160; cat $O/pcie.txt; cat $O/frontbench.txt | cut -c1-200; cat $O/hostpath.txt | cut -c1-200
