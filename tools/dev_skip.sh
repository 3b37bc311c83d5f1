#!/bin/bash
# Per-class sensitivity of the step and of the forward convs' in-step rate: bench.py
# on the DEVELOPMENT build (make -C rau_vqa_amd/csrc dev -> librau_dev.so) with every launch of the named
# kernel classes dropped.  TIMING ONLY: the numerics of such a run are meaningless.
# usage: bash tools/dev_skip.sh "" "enc_h2h_gemm,lstm_fwd" "att_fwd_fused" ...     (development tool)
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd "$R"
[ -f rau_vqa_amd/librau_dev.so ] || { echo "build it first: make -C rau_vqa_amd/csrc dev" >&2; exit 1; }
mkdir -p gpurun_out
out=gpurun_out/dev_skip.log
: > $out
run() { echo "== skip: ${1:-(nothing)}" >> $out
  RAU_LIB=$R/rau_vqa_amd/librau_dev.so RAU_DEV_SKIP="$1" timeout -k 10 240 python3 bench.py --steps 20 --warmup 5 --quick ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_classes_ms_per_step']
        print('ms_per_step %.3f' % d['ms_per_step'], 'bulk mfma_frac', {n: v['mfma_frac'] for n, v in d.get('bulk_kernels', {}).items()},
              'bulk ms', {n: v['ms_per_step'] for n, v in d.get('bulk_kernels', {}).items()})
" >> $out 2>&1; }
for rep in 1 2; do
  for cfg in "$@"; do run "$cfg"; done
done
cat $out
