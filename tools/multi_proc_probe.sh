cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/e2e.bin","wb").write(K.by_name("system", 64 << 20))
PY
make -s -C qat-zstd-plugin_amd/test benchmark >/dev/null 2>&1; BM=qat-zstd-plugin_amd/test/benchmark
echo "three level-1 processes and a level-3 one on one GPU:"
for i in 1 2 3; do (QZSTD_HIP_DEBUG=1 $BM -m1 -t4 -l8 -c131072 -L1 /tmp/e2e.bin 2>&1 | grep -E "wall clock|Producer errors|not answered|Latency" | cut -c1-230 | sed "s/^/  p$i: /" &) ; done
QZSTD_HIP_DEBUG=1 $BM -m1 -t4 -l8 -c131072 -L3 /tmp/e2e.bin 2>&1 | grep -E "wall clock|Producer errors|not answered|Latency" | cut -c1-230 | sed "s/^/  p4(L3): /"; sleep 6
