#!/bin/bash
# kernel trace of one bench run; prints per-kernel durations of the recurrence's small kernels
# usage (GPU box): bash tools/ktrace_chain.sh [bench args]      (development tool)
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sk
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sk -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs "$@" > $R/gpurun_out/sk.log 2>&1
cd $R && python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/sk/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if True:
        d[(n[:64], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(k, len(v), "sum %.0f avg %.1f med %.1f min %.1f" % (sum(v), sum(v) / len(v), v[len(v) // 2], v[0]))
print("total kernel time %.1f ms over %d kernels" % (sum(sum(v) for v in d.values()) / 1e3, sum(len(v) for v in d.values())))
PY
