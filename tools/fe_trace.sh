#!/bin/bash
# GPU box: kernel trace of the batch front-end — how long the announcements' launches take and how many run at a time
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/fe.bin","wb").write(K.system_corpus(512 << 20)[0])
PY
make -s -C qat-zstd-plugin_amd >/dev/null 2>&1; make -s -C qat-zstd-plugin_amd/test frontbench >/dev/null 2>&1
FB=$PWD/qat-zstd-plugin_amd/test/frontbench
for CFG in "${@:-4 2}"; do set -- $CFG; S=$1; A=$2; shift 2
rm -rf /tmp/fetr; mkdir -p /tmp/fetr
(cd /tmp && env QZSTD_FRONT_AHEAD=$A "$@" timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fetr -- $FB -t18 -l${LOOPS:-6} -c${CH:-131072} -L${LV:-1} -s$S -m1 /tmp/fe.bin 2>&1 | grep -o "median [0-9.]* min [0-9.]* max [0-9.]*")
python - "$S" "$A" "$*" <<PY
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/fetr/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)) for r in rows if "find_sequences" in r["Kernel_Name"]]
ev.sort()
n = len(ev)
if n:
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    dur = sorted((e[1] - e[0]) / 1e3 for e in ev)
    busy = 0; cur_end = ev[0][0]
    for s, e, _, _ in ev:
        if e > cur_end: busy += e - max(s, cur_end); cur_end = e
    conc = sum(e[1] - e[0] for e in ev) / max(busy, 1)
    wgs = [e[2] // max(e[3], 1) for e in ev]
    qs = {}
    for r in rows:
        if "find_sequences" in r["Kernel_Name"]:
            q = r.get("Queue_Id", "?")
            qs.setdefault(q, [0, 0])
            qs[q][0] += 1
            qs[q][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("  hardware queues used: %d; launches / busy %% per queue: %s" % (len(qs), ", ".join("%d/%.0f" % (v[0], 100.0 * v[1] / (t1 - t0)) for _, v in sorted(qs.items()))) if False else "", end="")
    print("  hardware queues used: %d; launches / busy %% of the run per queue: %s" % (len(qs), ", ".join("%d/%.0f" % (v[0], 100.0 * v[1] / max(ev[-1][1] - ev[0][0], 1)) for _, v in sorted(qs.items()))))
    print("seg %s MiB ahead %s %s: %d launches, workgroups per launch median %d; duration us P10 %.0f P50 %.0f P90 %.0f max %.0f; GPU busy %.0f %% of %.1f ms; launches running at a time (while busy) %.1f"
          % (sys.argv[1], sys.argv[2], sys.argv[3], n, sorted(wgs)[n // 2], dur[n // 10], dur[n // 2], dur[9 * n // 10], dur[-1], 100.0 * busy / (t1 - t0), (t1 - t0) / 1e6, conc))
PY
done
