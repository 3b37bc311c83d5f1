#!/bin/bash
# timelines of one step: B=256 training, B=256 evaluate-mode forward, B=64 MS training (development tool)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name B env...
  name=$1; B=$2; shift 2
  f=gpurun_out/tl_$name.csv
  env "$@" RAU_PROF_TIMELINE=$f python3 tools/tlrun.py $B 512 > gpurun_out/tl_$name.txt 2>&1
  python3 tools/tl3.py $f full >> gpurun_out/tl_$name.txt 2>&1
  head -16 gpurun_out/tl_$name.txt
}
run train256 256 X=0
run train256s 256 RAU_TL_SPARSE=1
run eval256 256 RAU_TL_MODE=eval
run ms64 64 RAU_TL_VARIANT=MS
