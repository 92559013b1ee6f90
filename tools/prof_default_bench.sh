#!/bin/bash
# GPU box: rocprofv3 --kernel-trace of the DEFAULT bench command's timed path (bench.py --no-cpu: the roofline launches + the end-to-end leg),
# the match-finder's launches grouped by grid size: the 8192-workgroup launches are the `roofline` block's, the small ones the front-end's
# announcements (same kernel name, so `--stats` alone would average the two kinds together).
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/dbt; mkdir -p /tmp/dbt
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dbt -o db -- python $R/bench.py --no-cpu > /tmp/dbt/run.log 2>&1)
python - <<PY
import csv, glob, json, collections, re
rows = []
for f in glob.glob("/tmp/dbt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
g = collections.defaultdict(list)
for r in rows:
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    m = re.search(r"(qzstd_\w+(?:<[^>]*>)?|__amd_\w+|\w+_kernel)", r["Kernel_Name"])
    g[(m.group(1) if m else r["Kernel_Name"][:60], "8192 workgroups" if wg == 8192 else ("<= 64 workgroups" if wg <= 64 else "other"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel | launches | grid | average us | min | max")
for (k, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print("%s | %d | %s | %.1f | %.1f | %.1f" % (k, len(v), grid, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
line = [l for l in open("/tmp/dbt/run.log") if l.startswith("{")]
if not line:
    print("".join(open("/tmp/dbt/run.log").readlines()[-15:]))
if line:
    d = json.loads(line[-1])
    print("bench line of this (traced) run: value %.1f MB/s, roofline.kernel_ms_avg %.3f (HIP events)" % (d["value"], d["roofline"]["kernel_ms_avg"]))
PY
