#!/bin/bash
# same-box A/B of the configs[2] workload (bf16 operands, D = 2048) under environment settings
# usage: bash tools/ab_bf16.sh "VAR=.." ...   (development tool; RAU_LIB=path A/Bs two builds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ab_bf16.log; : > $out
run() { echo "== $*" >> $out; timeout -k 10 240 env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --dtype bf16 --D 2048 --variant ResNet 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.3f' % d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['bulk_kernels'].items()})
" >> $out 2>&1; }
for rep in 1 2; do
  run X=0
  for cfg in "$@"; do run $cfg; done
done
cat $out
