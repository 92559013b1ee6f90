#!/bin/bash
# Memory-side PMC passes (TA / TCP / TCC) for one kernel shape (GPU box), few counters per pass, every pass under a short timeout.
# usage: tools/prof_mem.sh <tag> <shape of tools/ktime.py>
set -u
TAG=$1; SHAPE=${2:-6:131072:2048:system}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCC_BUSY_avr"; do
  i=$((i+1))
  KTIME_REPS=2 timeout -s KILL 100 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p$i -- python $R/tools/ktime.py $SHAPE > $OUT/p$i.log 2>&1 || echo "pass $i ($C): failed or timed out"
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qzstd_find" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(tot):
        line = "%-40s per-launch %18.1f  (launches %d)" % (k, tot[k] / max(n[k], 1), n[k])
        print(line); o.write(line + "\n")
PY
