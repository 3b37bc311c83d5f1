#!/bin/bash
# same-box A/B of one environment setting across the four workloads that matter (development tool):
# configs[1] f32 B=256, configs[2] bf16 D=2048, the configs[3] shard (B=64 MS), and the inference leg.
# usage: bash tools/ab_modes.sh "VAR=.. VAR=.."
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ab_modes.log
: > $out
run() { # label, bench args, env...
  label=$1; args=$2; shift 2
  echo "== $label [$*]" >> $out
  timeout -k 10 240 env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $args 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.3f  inference %.0f QA/s' % (d['ms_per_step'], d.get('inference_qa_per_s', 0)))
" >> $out 2>&1; }
for rep in 1 2; do
  for cfg in "X=0" "$@"; do
    run f32_256 "" $cfg
    run bf16_2048 "--dtype bf16 --D 2048 --variant ResNet" $cfg
    run ms_64 "--batch 64 --variant MS" $cfg
  done
done
cat $out
