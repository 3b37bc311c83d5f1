#!/bin/bash
# PMC passes over a stand-alone tool binary: bash tools/pmc_tool.sh <kernel-substring> <binary> [args]
# (FETCH_SIZE / WRITE_SIZE / SQ set in separate runs; prints per-kernel averages)   development tool
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
pat=$1; shift
bin=$R/tools/$1; shift
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmct_$tag
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmct_$tag -- $bin "$@" > $R/gpurun_out/pmct_$tag.log 2>&1
done
cd $R && python3 - "$pat" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT"):
    fs = glob.glob(f"gpurun_out/pmct_{tag}/**/*counter_collection.csv", recursive=True)
    if not fs: print(tag, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
