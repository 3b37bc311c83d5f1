"""Summarise tools/pmc_l2.sh: TCP->TCC (L2) requests per kernel and per step (development tool).

The request size is calibrated on a kernel whose traffic is known exactly: k_dropout_features_gen reads the
feature map once and writes H masked copies (bench.py defaults: B=256, D=512, S=196, H=8), nothing else.
usage: python tools/pmc_l2_summary.py gpurun_out/pmc_l2 [steps_profiled]"""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
B, D, S, H = 256, 512, 196, 8
files = sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv")), key=os.path.getmtime)
if not files:
    raise SystemExit("no counter_collection.csv under " + src)
rows = list(csv.DictReader(open(files[-1])))


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("rau::", "")


per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = short(r["Kernel_Name"])
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
drop = next((k for k in per if k.startswith("k_dropout_features_gen")), None)
calib = None
if drop:
    n = len(disp[drop])
    rd = per[drop]["TCP_TCC_READ_REQ_sum"] / n
    wr = per[drop]["TCP_TCC_WRITE_REQ_sum"] / n
    calib = (B * D * S * 4.0 / rd if rd else None, H * B * D * S * 4.0 / wr if wr else None)
steps = None
if drop:
    steps = len(disp[drop])
print("# L2 request counters per kernel (`bash tools/pmc_l2.sh`, kernels serialised by the profiler)\n")
if calib:
    print(f"Calibration on `{drop}` (reads X = {B * D * S * 4 / 1e6:.0f} MB, writes {H} copies = "
          f"{H * B * D * S * 4 / 1e6:.0f} MB per launch, {steps} launches): "
          f"{calib[0]:.1f} bytes per TCP_TCC_READ_REQ, {calib[1]:.1f} bytes per TCP_TCC_WRITE_REQ.\n")
rb = calib[0] if calib and calib[0] else 64.0
wb = calib[1] if calib and calib[1] else 64.0
print("| kernel | launches | read MB / launch | write MB / launch | read GB / step | L2 hit rate |")
print("|---|---|---|---|---|---|")
tot_r = tot_w = 0.0
for k, c in sorted(per.items(), key=lambda kv: -kv[1]["TCP_TCC_READ_REQ_sum"]):
    n = len(disp[k])
    r_mb = c["TCP_TCC_READ_REQ_sum"] * rb / 1e6
    w_mb = c["TCP_TCC_WRITE_REQ_sum"] * wb / 1e6
    tot_r += r_mb
    tot_w += w_mb
    if r_mb / max(steps or 1, 1) < 20:
        continue
    hit = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
    print(f"| `{k}` | {n} | {r_mb / n:.1f} | {w_mb / n:.1f} | {r_mb / 1e3 / max(steps or 1, 1):.2f} | {hit:.2f} |")
if steps:
    print(f"\nAll kernels of the run ({steps} steps incl. warm-up): {tot_r / 1e3 / steps:.1f} GB read and "
          f"{tot_w / 1e3 / steps:.1f} GB written through L2 per step (TCP -> TCC requests x the calibrated size).")
