#!/bin/bash
# development probe: kernel stats of the configs[2] step with and without bf16 products in the skinny GEMMs
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
cd "$R"
A="--dtype bf16 --D 2048 --variant ResNet --quick --steps 20 --warmup 3"
rm -rf gpurun_out/skbf0 gpurun_out/skbf1
export RAU_SKINNY_BF16=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/skbf0 -- python3 bench.py $A > gpurun_out/skbf0.log 2>&1; echo "rc=$?"
export RAU_SKINNY_BF16=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/skbf1 -- python3 bench.py $A > gpurun_out/skbf1.log 2>&1; echo "rc=$?"
for d in skbf0 skbf1; do echo "== $d"; f=$(ls gpurun_out/$d/*/*_kernel_stats.csv | head -1); head -16 "$f" | cut -d, -f1-5 | cut -c1-150; done
