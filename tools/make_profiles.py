"""Turn one round's rocprofv3 outputs into the committed profiles/ artefacts (development tool).

Inputs (written on the GPU box by tools/profile_round.sh into gpurun_out/):
  kt/      rocprofv3 --kernel-trace --stats           -- python3 bench.py --steps 20 --warmup 3
  pmc_f/   rocprofv3 --pmc FETCH_SIZE --kernel-trace  -- python3 bench.py --steps 2 --warmup 1
  pmc_w/   rocprofv3 --pmc WRITE_SIZE --kernel-trace  -- same
  pmc_sq/  rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE       -- same
  bench_line.json   the JSON line of an un-profiled `python bench.py` on the same box
Outputs: profiles/<tag>_kernel_stats.csv, <tag>_pmc_<class>.json, <tag>_sq_counters.md,
<tag>_bench_line.json, <tag>_summary.md.

usage: python tools/make_profiles.py r02 gpurun_out
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag, src = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
B, S, H = 256, 196, 8          # bench.py defaults (configs[1])
D, M, A = 512, 512, 256


def newest(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    if not files:
        raise SystemExit("missing " + pattern)
    return files[-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("rau::", "")


def classify(rows):
    """Label the bulk conv dispatches (kernel template + grid + neighbours in dispatch order)."""
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    grids = [int(r["Grid_Size"]) // 256 for r in rows]
    out = [None] * len(rows)
    # round 3: the forward convs run on the wide tiling (conv_wide.hip, k_conv_wide<0>); the
    # flattened-column kernel only takes what the wide one does not (other map sizes, remainders)
    fwd = "gemm_kernel<128, 128, 32, 1, 2, 2, 0>"
    fwd_idx = [i for i, n in enumerate(names) if n == fwd or n.startswith("k_conv_wide<0>")]
    for pos, i in enumerate(fwd_idx):
        g = grids[i]
        # in stream order every conv_embed_fwd launch is followed by its conv_att_pre launch
        # (half the workgroups: A = M / 2); dispatch ids interleave with other streams, so pair
        # up consecutive launches of this kernel name instead
        if pos % 2 == 0:
            out[i] = ("conv_embed_fwd", None)
        else:
            out[i] = ("conv_att_pre", None)
    # conv weight gradients: with the tanh factor applied in the dgrad's epilogue both use the
    # plain (SC, SC) kernel; in stream order conv_att_wgrad precedes conv_embed_wgrad in every group
    # (round 3: wgrad_dma.hip's k_wgrad_dma takes them on 14 x 14 maps)
    plain = [i for i, n in enumerate(names) if "128, 128, 28, 3, 3, 1, 0>" in n or "k_wgrad_dma" in n]
    dtanh = [i for i, n in enumerate(names) if "128, 128, 28, 4, 3, 1, 0>" in n]
    if dtanh:
        for i in plain:
            out[i] = ("conv_att_wgrad", None)
        for i in dtanh:
            out[i] = ("conv_embed_wgrad", None)
    else:
        for pos, i in enumerate(plain):
            out[i] = ("conv_att_wgrad" if pos % 2 == 0 else "conv_embed_wgrad", None)
    for i, n in enumerate(names):
        if n.startswith("k_conv_sample<1") or n.startswith("k_conv_sample<2") or n.startswith("k_dgrad_dma") or \
                n.startswith("k_conv_wide<2>") or n == "gemm_kernel<128, 128, 32, 1, 2, 3, 0>":
            out[i] = ("conv_att_dgrad", None)
    return out


def counter_rows(d, name):
    f = newest(os.path.join(d, "*", "*_counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    return rows, f


# ---- 1. kernel stats
stats = newest("kt/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(OUT, f"{tag}_kernel_stats.csv"))
trace = list(csv.DictReader(open(newest("kt/*/*_kernel_trace.csv"))))

# ---- 2. HBM traffic per launch of each bulk class
fr, ffile = counter_rows("pmc_f", "FETCH_SIZE")
wr, wfile = counter_rows("pmc_w", "WRITE_SIZE")
traffic = {}
for rows, key in ((fr, "fetch"), (wr, "write")):
    lab = classify(rows)
    for r, l in zip(rows, lab):
        if l is None:
            continue
        wg = int(r["Grid_Size"]) // 256
        e = traffic.setdefault(l[0], {}).setdefault(wg, {"fetch": [], "write": []})
        e[key].append(float(r["Counter_Value"]))
# algorithmic bytes per hop (f32): operands read once + result written once
alg_per_hop = {
    "conv_embed_fwd": (B * D * S + B * M * S) * 4 + M * D * 4,
    "conv_att_pre": (B * M * S + B * A * S) * 4 + A * M * 4,
    "conv_att_dgrad": (B * A * S + 2 * B * M * S) * 4 + A * M * 4,   # reads dS and I, writes dZ
    "conv_att_wgrad": (B * A * S + B * M * S) * 4,
    "conv_embed_wgrad": (B * M * S + B * D * S) * 4,                  # reads dZ and X
}
# workgroups per hop where the grid scales with the hops of a launch (else: 1.6 hops per launch
# on average with hop groups 2,2,2,1,1)
wg_per_hop = {"conv_embed_fwd": 512, "conv_att_pre": 256, "conv_att_dgrad": 1024}   # wide tiles: (M/64) x (B/4) per hop
pmc_json = {}
for cls, by_wg in traffic.items():
    launches = []
    for wg, e in sorted(by_wg.items()):
        if not e["fetch"] or not e["write"]:
            continue
        f = sum(e["fetch"]) / len(e["fetch"])
        w = sum(e["write"]) / len(e["write"])
        launches.append({"workgroups": wg, "dispatches_fetch_pass": len(e["fetch"]),
                         "dispatches_write_pass": len(e["write"]),
                         "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w,
                         "hbm_bytes": (2 * f + w) * 1024})
    if not launches:
        continue
    n = sum(l["dispatches_fetch_pass"] for l in launches)
    avg = sum(l["hbm_bytes"] * l["dispatches_fetch_pass"] for l in launches) / n
    if cls in wg_per_hop:
        for l in launches:
            l["algorithmic_bytes"] = alg_per_hop[cls] * l["workgroups"] / wg_per_hop[cls]
        alg = sum(l["algorithmic_bytes"] * l["dispatches_fetch_pass"] for l in launches) / n
    else:
        alg = alg_per_hop[cls] * H / 5.0   # five launches per step (hop groups 2,2,2,1,1)
    pmc_json[cls] = {
        "kernel": cls,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- "
                   "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs (two separate passes)",
        "launch_shapes": launches,
        "hbm_bytes_per_launch": avg,
        "algorithmic_bytes_per_launch": alg,
        "ratio": avg / alg,
        "note": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads). Average over "
                "the launches of a step weighted by dispatch count.",
    }
    json.dump(pmc_json[cls], open(os.path.join(OUT, f"{tag}_pmc_{cls}.json"), "w"), indent=1)

# ---- 3. SQ counters
sqf = newest("pmc_sq/*/*_counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
ns = collections.defaultdict(float)
for r in csv.DictReader(open(sqf)):
    k = short(r["Kernel_Name"])[:64] + f" wg={int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)}"
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in cnt[k]:
        ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k].add(r["Dispatch_Id"])
    agg[k]["_vgpr"] = float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"])
    agg[k]["_lds"] = float(r["LDS_Block_Size"])
lines = ["# %s: SQ counters per kernel (rocprofv3 --pmc, kernels serialised by the profiler)" % tag, "",
         "`MfmaBusy` = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): fraction of the four matrix",
         "pipes' cycles that executed an MFMA while the CU was busy (ROCm 7.2 ships no gfx950 derived-metric",
         "section, so MfmaUtil is formed by hand).  wait/issue/active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY /",
         "SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (parked on s_waitcnt or barrier / stalled at issue, i.e.",
         "matrix pipe busy / issuing).  LDSconf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  clk = ",
         "GRBM_GUI_ACTIVE / 8 / duration.  Occupancy limiter: VGPRs (+AGPRs) and LDS bytes per workgroup.", "",
         "| kernel | n | avg us | VGPR | LDS B | MfmaBusy | wait | issue-stall | active | LDSconf | clk GHz |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
rows = []
for k, v in agg.items():
    n = len(cnt[k])
    us = ns[k] / n / 1e3
    if us * n < 300:
        continue
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    rows.append((us * n, "| `%s` | %d | %.1f | %d | %d | %.2f | %.2f | %.2f | %.2f | %.3f | %.2f |" % (
        k, n, us, v["_vgpr"], v["_lds"],
        v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4 * v.get("SQ_BUSY_CU_CYCLES", 1), 1),
        v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc,
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        v.get("GRBM_GUI_ACTIVE", 0) / 8 / max(ns[k], 1))))
lines += [r for _, r in sorted(rows, reverse=True)[:30]]
open(os.path.join(OUT, f"{tag}_sq_counters.md"), "w").write("\n".join(lines) + "\n")

# ---- 4. bench line + summary
bl = os.path.join(src, "bench_line.json")
line = None
if os.path.exists(bl):
    for l in open(bl):
        if l.startswith("{"):
            line = json.loads(l)
    if line:
        json.dump(line, open(os.path.join(OUT, f"{tag}_bench_line.json"), "w"), indent=1)
by = collections.defaultdict(lambda: [0, 0.0])
for r in trace:
    k = short(r["Kernel_Name"])[:64] + f" wg={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}"
    by[k][0] += 1
    by[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
top = sorted(by.items(), key=lambda kv: -kv[1][1])[:14]
sm = ["# %s summary" % tag, ""]
if line:
    rf = line["roofline"]
    sm += ["* un-profiled `python bench.py`: %.2f ms/step, %.0f QA-pairs/s; dominant bulk kernel `%s`: %.1f TFLOP/s = %.2f of %s peak "
           "(HIP events, avg %.1f us over %.0f launches/step); cpu_baseline %.0f QA/s on %d cores." % (
               line["ms_per_step"], line["value"], rf["kernel"], rf["mfma_tflops"], rf["mfma_frac"],
               "f32 MFMA", rf["avg_launch_ms"] * 1e3, rf["launches_per_step"],
               line.get("cpu_baseline", {}).get("value", float("nan")),
               line.get("cpu_baseline", {}).get("cores", 0)), ""]
sm += ["Kernel-trace averages (`%s_kernel_stats.csv`, same command with `--steps 20 --warmup 3`):" % tag, "",
       "| kernel (workgroups) | calls | avg us |", "|---|---|---|"]
sm += ["| `%s` | %d | %.1f |" % (k, n, t / n / 1e3) for k, (n, t) in top]
sm += ["", "HBM traffic per launch (separate `--pmc` passes, `%s_pmc_*.json`):" % tag, "",
       "| kernel | measured MB | algorithmic MB | ratio |", "|---|---|---|---|"]
sm += ["| %s | %.1f | %.1f | %.2f |" % (c, j["hbm_bytes_per_launch"] / 1e6,
                                       j["algorithmic_bytes_per_launch"] / 1e6, j["ratio"])
       for c, j in sorted(pmc_json.items())]
open(os.path.join(OUT, f"{tag}_summary.md"), "w").write("\n".join(sm) + "\n")
print("\n".join(sm))
