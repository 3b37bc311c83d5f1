#!/bin/bash
# timelines of one step under different RAU_CONV_WIDE settings (development tool)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "0 2" "7 2" "7 1"; do
  set -- $cfg
  f=gpurun_out/tl_w$1_c$2.csv
  RAU_CONV_WIDE=$1 RAU_CONV_WIDE_PER_CU=$2 RAU_PROF_TIMELINE=$f python3 tools/tlrun.py 256 512 > gpurun_out/tl_w$1_c$2.txt 2>&1
  python3 tools/tl3.py $f full >> gpurun_out/tl_w$1_c$2.txt 2>&1
  head -20 gpurun_out/tl_w$1_c$2.txt
done
