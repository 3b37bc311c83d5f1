#!/bin/bash
# the attention kernels (register-staged, LDS-DMA with 8 / 16 waves) against a host reference (development tool)
R="$(cd "$(dirname "$0")/.." && pwd)"
[ -n "$R" ] && [ -f "$R/bench.py" ] || { echo "cannot locate the repo root from $0" >&2; exit 1; }
cd "$R"
mkdir -p gpurun_out
for cfg in "8 40 20 12" "8 136 132 196" "16 512 256 196" "6 40 20 52"; do
  echo "== $cfg"; echo -n "old: "; RAU_ATT_DMA_OFF=1 ./tools/attcheck $cfg gpurun_out/att_ref.bin
  echo -n "dma8: "; ./tools/attcheck $cfg gpurun_out/att_new.bin
  echo -n "dma16: "; RAU_ATT_WAVES_FWD=16 ./tools/attcheck $cfg gpurun_out/att_new.bin
done
