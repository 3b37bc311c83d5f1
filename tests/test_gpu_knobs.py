"""Every A/B environment variable librau.so still reads keeps the parity bar:
each setting runs tests/knob_check.py -- train and evaluate mode against the fp64 oracle, 1e-4 -- in
its own process, because most of them are read once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (environment, batch, dims of tests/knob_check.py: "" generic, "ws" Rq = 512 (reaches enc_ws.hip),
#  "wg" A = M = D = 128 (reaches wgrad_dma.hip), "bf16" M = D = 256 (reaches wgrad16.hip / dgrad16.hip),
#  "bf16ws" bf16 mode at Rq = 512)
KNOBS = [
    ({}, 72, ""), ({}, 24, ""),                                 # defaults on both sides of the 64-sample switch
    ({"RAU_CONV_WIDE": "0"}, 72, ""),                           # round-2 tilings everywhere
    ({"RAU_CONV_WIDE_PER_CU": "2"}, 72, ""),                    # two wide workgroups per CU
    ({"RAU_CONV_WIDE": "0", "RAU_CONV_SAMPLE": "15"}, 72, ""),  # per-sample tiling for all four convs
    ({"RAU_CONV_WIDE": "0", "RAU_CONV_SAMPLE": "0"}, 72, ""),   # flattened-column tiling for all
    ({"RAU_HOP_GROUPS": "1,3", "RAU_BWD_GROUPS": "2,2"}, 72, ""),
    ({"RAU_ATT_WAVES_FWD": "16", "RAU_ATT_WAVES_BWD": "8"}, 72, ""),
    ({"RAU_ATT_DMA_OFF": "1"}, 72, ""),
    ({}, 72, "wg"),                                             # LDS-DMA f32 conv weight gradients on ...
    ({"RAU_WGRAD_DMA_OFF": "1"}, 72, "wg"),                     # ... and off (register-staged tile)
    ({"RAU_DGRAD_DMA": "0"}, 72, "wg"),                         # f32 attention dgrad: register-staged per-sample kernel
    ({"RAU_DGRAD_DMA": "2"}, 72, "wg"),                         # ... dgrad_dma.hip with a ring of two stages (default three)
    ({"RAU_WG_BULK": "0"}, 72, "wg"),                           # grouped Linear weight gradients on the third stream ...
    ({"RAU_WG_BULK": "3"}, 72, "wg"),                           # ... and both groups at the end of the bulk stream
    ({"RAU_WGRAD_GROUPS": "1"}, 72, "wg"),                      # conv weight gradients: four-wave workgroups for both ...
    ({"RAU_WGRAD_GROUPS": "2"}, 72, "wg"),                      # ... and the eight-wave (two K groups) form for both
    ({"RAU_SKINNY_DMA_OFF": "1"}, 72, ""),
    ({"RAU_SKINNY_ALIGN16": "1"}, 72, ""),                      # weights off a 16-byte boundary back on the register-staged tile
    ({"RAU_SKINNY_RAGGED_OFF": "1"}, 72, ""),                   # ... and the K = 196 / N = 196 products (skinny_dma.hip RAG)
    ({"RAU_SKINNY_RAGGED_OFF": "1"}, 24, ""),
    ({"BF16": "1"}, 12, "bf16"),                                # bf16 mode, wgrad16.hip on ...
    ({"BF16": "1", "RAU_WGRAD16_OFF": "1"}, 12, "bf16"),        # ... and off (round-2 tile)
    ({"BF16": "1", "RAU_SKINNY_DMA_OFF": "1"}, 12, "bf16"),     # ... its Linear products on the register-staged tiles (rounded while staged)
    ({"BF16": "1"}, 32, "bf16ws"),                              # ... and through the persistent encoder's rounding form
    ({"RAU_SEAM": "0"}, 72, ""),                                # head dgrad in the backward, conv gradients behind the hop's last launch
    ({"RAU_SEAM": "1"}, 72, ""), ({"RAU_SEAM": "2"}, 72, ""),   # ... each half of the default (3) alone
    ({"RAU_ATT_SPLIT": "1", "RAU_ATT_CHUNKS": "4"}, 72, ""),
    ({"RAU_ATT_FUSED": "1"}, 24, ""),
    ({}, 32, "ws"),                                             # persistent encoder selected by shape ...
    ({"RAU_ENC_WS": "0"}, 32, "ws"),                            # ... and forced off at the same shape
    ({"RAU_ENC_WS": "1"}, 80, "ws"),                            # ... and forced on where it is not chosen
    ({"RAU_SKINNY_DEEP": "1"}, 72, ""),                         # 32-deep stages for the skinny GEMMs at any shape
    ({"RAU_SKINNY_DEEP": "0"}, 24, ""),                         # ... and 16-deep where 32 would be chosen
    ({"RAU_SIDE_SPLIT": "1"}, 72, ""),                          # the chain's non-recurrent GEMMs on the side stream
    ({"RAU_SIDE_SPLIT": "0"}, 24, ""),                          # ... and kept on the chain where they would be split
    ({"RAU_ENC_CHUNKS": "1"}, 24, ""),                          # the side stream's layer-1 input projection as one launch (default: four chunks)
]


@pytest.mark.parametrize("env,batch,kind", KNOBS,
                         ids=[" ".join(f"{k}={v}" for k, v in e.items()) + f" B={b} {kd}".rstrip()
                              for e, b, kd in KNOBS])
def test_knob_setting_keeps_parity(env, batch, kind):
    e = dict(os.environ, **env)
    args = [str(batch)] + ([kind] if kind else [])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "knob_check.py")] + args,
                         env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
