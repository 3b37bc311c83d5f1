#!/bin/bash
# L2-side request counters of one bench run (the L2 -> CU path: what LDS-DMA and global loads pull through the
# TCP), kernels serialised by the profiler.  The program sits directly behind `--` (python3).
# usage: bash tools/pmc_l2.sh ; python tools/pmc_l2_summary.py gpurun_out/pmc_l2 > profiles/rNN_l2_requests.md
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
cd "$R"
rm -rf gpurun_out/pmc_l2
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_l2 -- python3 bench.py --steps 2 --warmup 1 --quick > gpurun_out/pmc_l2.log 2>&1; echo "l2 rc=$?"
ls gpurun_out/pmc_l2/*/ | head
