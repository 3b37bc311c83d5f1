# A/B of ROCm's graph-executor knob on the hipGraph step (development tool)
run() { python bench.py "$@" --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for cfg in "--batch 128 --D 2048 --dtype bf16 --variant Full" "--batch 64 --variant MS" "--batch 256"; do
  echo "== $cfg"
  echo eager; run $cfg
  echo graph default; run $cfg --graph
  for q in 2 8; do echo graph FORCE_GRAPH_QUEUES=$q; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q run $cfg --graph; done
done
