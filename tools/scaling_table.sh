#!/bin/bash
# Strong-scaling projection table of DESIGN.md section 7: whole batch on one GPU vs one rank's shard at N = 8
# (eager, no all-reduce), plus the hipGraph replay of the shards and the f32 D = 2048 step (development tool).
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd "$R"
mkdir -p gpurun_out
out=gpurun_out/scaling_table.log
: > $out
run() { echo "== $*" >> $out; timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.3f  value %.0f' % (d['ms_per_step'], d['value']))
" >> $out 2>&1; }
run --batch 512 --variant MS
run --batch 64 --variant MS
run --batch 64 --variant MS --graph
run --batch 1024 --D 2048 --variant Full --dtype bf16
run --batch 128 --D 2048 --variant Full --dtype bf16
run --batch 128 --D 2048 --variant Full --dtype bf16 --graph
run --batch 1024 --D 2048 --variant Full
run --batch 128 --D 2048 --variant Full
run --batch 256 --D 2048 --variant ResNet
cat $out
