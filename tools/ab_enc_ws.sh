#!/bin/bash
# the weight-stationary encoder on / off at B = 16 / 32 / 64: training step and inference (development tool)
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd "$R"
mkdir -p gpurun_out
for B in 16 32 64; do
for ws in 1 0; do
RAU_ENC_WS=$ws python3 bench.py --batch $B --variant MS --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_classes_ms_per_step']
        print('B=$B ws=$ws ms_per_step %.3f' % d['ms_per_step'], {n: k[n] for n in k if n.startswith('enc_ws') or n.startswith('lstm_fwd') or n == 'enc_h2h_gemm'}, 'inference', round(d['inference_qa_per_s']))
"
done; done
