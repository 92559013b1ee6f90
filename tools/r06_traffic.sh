#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the bench's roofline kernel (two rocprofv3 --pmc passes, nothing else in them): usage tools/r06_traffic.sh <tag> [library]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; SO=${2:-}
OUT=$R/gpurun_out/traffic_$TAG; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  QZ_PLUGIN_SO=$SO timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o $C -- python $R/bench.py --steps 2 --warmup 1 --kernel-only > $OUT/$C.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qzstd_find" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-12s per-launch %16.1f KiB (launches %d)" % (k, tot[k] / max(n[k], 1), n[k]))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    b = (2 * tot["FETCH_SIZE"] / n["FETCH_SIZE"] + tot["WRITE_SIZE"] / n["WRITE_SIZE"]) * 1024
    print("2 x FETCH + WRITE = %.3f GB per launch = %.2f x the algorithmic 1.924 GB" % (b / 1e9, b / 1923644224))
PY
