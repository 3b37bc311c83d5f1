#!/bin/bash
# Runs on the GPU box (gpurun): un-profiled bench line, kernel trace + stats, PMC passes.
# The program sits directly behind `--` (python3), as the pool requires.
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
cd "$R"
rm -rf gpurun_out/kt gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_sq
python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/kt.log 2>&1; echo "kt rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_f.log 2>&1; echo "f rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_w.log 2>&1; echo "w rc=$?"
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_sq.log 2>&1; echo "sq rc=$?"
grep -a '^{' gpurun_out/kt.log | head -1 | cut -c1-200
