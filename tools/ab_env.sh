#!/bin/bash
# same-box A/B of bench.py under environment settings: each argument is one "VAR=.. VAR=.." set
# usage: bash tools/ab_env.sh "RAU_A=1" "RAU_A=2 RAU_B=1" ...   (development tool; two interleaved rounds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ab_env.log
: > $out
run() { echo "== $*" >> $out; timeout -k 10 240 env $* python3 bench.py --steps 20 --warmup 5 --quick ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.3f' % d['ms_per_step'], {k: v['mfma_frac'] for k, v in d.get('bulk_kernels', {}).items()})
" >> $out 2>&1; }
for rep in 1 2; do
  run DUMMY=0
  for cfg in "$@"; do run $cfg; done
done
cat $out
