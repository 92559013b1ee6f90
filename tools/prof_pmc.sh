#!/bin/bash
# PMC passes for the match-finder kernel (run on the GPU box via gpurun).  Counters are collected
# in their own runs (no --kernel-trace/--stats mixed in), as MI355X_MICROARCH.md prescribes.
# usage: tools/prof_pmc.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --kernel-only "$@" > $OUT/p$i.log 2>&1
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
        line = "%-28s per-launch %18.1f  (launches %d)" % (k, tot[k] / max(n[k], 1), n[k])
        print(line); o.write(line + "\n")
PY
