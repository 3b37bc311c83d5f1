#!/bin/bash
# Same as profile_round.sh for the configs[2] workload (bf16 operands, D=2048): bench line, kernel trace,
# SQ counters and HBM traffic.  The program sits directly behind `--` (python3).
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
cd "$R"
A="--dtype bf16 --D 2048 --variant ResNet --no-cpu-baseline --no-other-configs"
rm -rf gpurun_out/b16_kt gpurun_out/b16_f gpurun_out/b16_w gpurun_out/b16_sq
python3 bench.py $A > gpurun_out/b16_bench_line.json 2> gpurun_out/b16_bench_line.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b16_kt -- python3 bench.py $A --steps 20 --warmup 3 > gpurun_out/b16_kt.log 2>&1; echo "kt rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/b16_f -- python3 bench.py $A --steps 2 --warmup 1 > gpurun_out/b16_f.log 2>&1; echo "f rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/b16_w -- python3 bench.py $A --steps 2 --warmup 1 > gpurun_out/b16_w.log 2>&1; echo "w rc=$?"
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/b16_sq -- python3 bench.py $A --steps 2 --warmup 1 > gpurun_out/b16_sq.log 2>&1; echo "sq rc=$?"
cut -c1-300 gpurun_out/b16_bench_line.json
