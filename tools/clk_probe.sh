#!/bin/bash
# shader clock and package power sampled while the bench loop runs (development tool)
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd "$R"
mkdir -p gpurun_out
python3 bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 22
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Socket" | tr '\n' ' '; echo; sleep 1; done
wait $BP
echo "--- idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|Socket" | tr '\n' ' '; echo
