#!/bin/bash
# A/B of the wide conv tiling inside the step (same box): RAU_CONV_WIDE mask x workgroups per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ab_wide.log
: > $out
run() { echo "== $*" >> $out; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'value', d['value'], {k: (round(v['mfma_frac'], 3), round(v.get('avg_us', 0), 1)) for k, v in d.get('bulk_kernels', {}).items()})
" >> $out 2>&1; }
for rep in 1 2; do
run RAU_CONV_WIDE=0
run RAU_CONV_WIDE=7 RAU_CONV_WIDE_PER_CU=2
run RAU_CONV_WIDE=7 RAU_CONV_WIDE_PER_CU=1
run RAU_CONV_WIDE=3 RAU_CONV_WIDE_PER_CU=2
run RAU_CONV_WIDE=3 RAU_CONV_WIDE_PER_CU=1
done
cat $out
