#!/bin/bash
# sparse timelines (bulk + weight-gradient streams bracketed, the recurrence at its un-profiled pace) of
# one step: configs[1] f32, configs[2] bf16 D=2048, the configs[3] shard (B=64 MS); full timeline of an
# evaluate-mode forward (development tool)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name B D env...
  name=$1; B=$2; D=$3; shift 3
  f=gpurun_out/tl_$name.csv
  rm -f $f
  env "$@" RAU_PROF_TIMELINE=$f python3 tools/tlrun.py $B $D > gpurun_out/tl_$name.txt 2>&1
  python3 tools/tl3.py $f full >> gpurun_out/tl_$name.txt 2>&1
}
run train256s 256 512 RAU_TL_SPARSE=1
run bf16s 256 2048 RAU_TL_SPARSE=1 RAU_TL_DTYPE=bf16 RAU_TL_VARIANT=ResNet
run ms64s 64 512 RAU_TL_SPARSE=1 RAU_TL_VARIANT=MS
run eval256 256 512 RAU_TL_MODE=eval
