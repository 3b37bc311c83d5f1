"""profiles/<tag>_bf16_* from tools/profile_bf16.sh's outputs (configs[2]: bf16 operands, D=2048).
usage: python tools/make_profiles_bf16.py r02 gpurun_out"""
import collections, csv, glob, json, os, re, shutil, sys
tag, src = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
B, S, H, D, M, A = 256, 196, 8, 2048, 512, 256
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("rau::", "")
def newest(p):
    f = sorted(glob.glob(os.path.join(src, p)), key=os.path.getmtime)
    if not f: raise SystemExit("missing " + p)
    return f[-1]
# kernel -> (class, algorithmic bytes per hop): bf16 operands where they are stored as bf16
CLASSES = {
    "gemm_kernel<128, 128, 32, 8, 6, 2, 1>": ("conv_embed_fwd", B * D * S * 2 + B * M * S * 4 + M * D * 2, 1568),
    "gemm_kernel<128, 128, 32, 8, 2, 2, 1>": ("conv_att_pre", (B * M * S + B * A * S) * 4 + A * M * 2, 784),
    "k_conv_sample<2, 2>": ("conv_att_dgrad", B * A * S * 4 + B * M * S * 4 + B * M * S * 2 + A * M * 4, 1024),
    "gemm_split_xcd_kernel<128, 128, 32, 3, 3, 1, 1>": ("conv_att_wgrad", (B * A * S + B * M * S) * 4, None),
    "gemm_split_xcd_kernel<128, 128, 32, 7, 7, 1, 1>": ("conv_embed_wgrad", B * M * S * 2 + B * D * S * 2, None),
    "k_wgrad16": ("conv_embed_wgrad", B * M * S * 2 + B * D * S * 2, None),   # round 3 (wgrad16.hip)
    # round 3, second half: dgrad16.hip (dS read as bf16, I f32, dZ written as bf16) and the att_i
    # weight gradient with dS stored as bf16
    "k_dgrad16<true>": ("conv_att_dgrad", B * A * S * 2 + B * M * S * 4 + B * M * S * 2 + A * M * 4, 1024),
    "gemm_split_xcd_kernel<128, 128, 32, 7, 3, 1, 1>": ("conv_att_wgrad", B * A * S * 2 + B * M * S * 4, None),
    "k_dropout_features_b16": ("dropout_features", None, None),
    "k_dropout_features_gen<true>": ("dropout_features", None, None),
}
def per_kernel(d, counter):
    f = newest(os.path.join(d, "*", "*_counter_collection.csv"))
    out = collections.defaultdict(list)
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if r["Counter_Name"] == counter:
            out[k].append(float(r["Counter_Value"]))
    return out, dur
fetch, dur = per_kernel("b16_f", "FETCH_SIZE")
write, _ = per_kernel("b16_w", "WRITE_SIZE")
rows = []
for (name, wg), f in fetch.items():
    if name not in CLASSES: continue
    cls, alg_hop, wg_hop = CLASSES[name]
    w = write.get((name, wg), [0])
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    hbm = (2 * fk + wk) * 1024
    us = sum(dur[(name, wg)].values()) / len(dur[(name, wg)])
    if alg_hop is None:   # dropout: reads X once, writes H bf16 copies
        alg = B * D * S * 4 + H * B * D * S * 2
    else:
        alg = alg_hop * (wg / wg_hop if wg_hop else H / 5.0)
    rows.append((cls, name, wg, len(f), us, hbm, alg))
rows.sort()
# per class: launch-count-weighted averages, the form bench.py reads (roofline.traffic)
by_cls = collections.defaultdict(list)
for r in rows: by_cls[r[0]].append(r)
for cls, rr in by_cls.items():
    n = sum(r[3] for r in rr)
    json.dump({"kernel": cls,
               "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 "
                          "bench.py --dtype bf16 --D 2048 --variant ResNet --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 "
                          "(two separate passes)",
               "launch_shapes": [{"kernel_name": r[1], "workgroups": r[2], "dispatches": r[3], "avg_us": r[4],
                                  "hbm_bytes": r[5], "algorithmic_bytes": r[6]} for r in rr],
               "hbm_bytes_per_launch": sum(r[5] * r[3] for r in rr) / n,
               "algorithmic_bytes_per_launch": sum(r[6] * r[3] for r in rr) / n,
               "ratio": sum(r[5] * r[3] for r in rr) / sum(r[6] * r[3] for r in rr),
               "note": "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction, "
                       "MI355X_MICROARCH.md); operands counted at their stored width"},
              open(os.path.join(OUT, f"{tag}_bf16_pmc_{cls}.json"), "w"), indent=1)
lines = [f"# {tag} bf16 mode (BASELINE.json configs[2]: Ours_ResNet, B=256, D=2048, bf16-rounded operands in every conv and Linear GEMM)", ""]
bl = os.path.join(src, "b16_bench_line.json")
for l in open(bl):
    if l.startswith("{"):
        line = json.loads(l)
        json.dump(line, open(os.path.join(OUT, f"{tag}_bf16_bench_line.json"), "w"), indent=1)
        lines += [f"* un-profiled `python bench.py --dtype bf16 --D 2048 --variant ResNet`: "
                  f"{line['ms_per_step']:.2f} ms/step, {line['value']:.0f} QA-pairs/s; roofline object: "
                  f"`{json.dumps(line.get('roofline'))}`", ""]
lines += ["HBM traffic of the bulk kernels, one launch each, kernels serialised by the profiler (separate",
          "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; hbm = (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950",
          "correction of MI355X_MICROARCH.md).  Algorithmic bytes count operands at their STORED width (xd, dZ and",
          "the transposed weights are bf16 in HBM in this mode; I, P, dS stay f32).", "",
          "| class | kernel | workgroups | launches | avg us | HBM MB | algorithmic MB | ratio | algorithmic TB/s |",
          "|---|---|---|---|---|---|---|---|---|"]
for cls, name, wg, n, us, hbm, alg in rows:
    lines.append(f"| {cls} | `{name}` | {wg} | {n} | {us:.1f} | {hbm / 1e6:.1f} | {alg / 1e6:.1f} | "
                 f"{hbm / alg:.2f} | {alg / us / 1e6:.2f} |")
# SQ counters
sqf = newest("b16_sq/*/*_counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ds = collections.defaultdict(dict)
for r in csv.DictReader(open(sqf)):
    k = short(r["Kernel_Name"])[:60] + f" wg={int(r['Grid_Size']) // int(r['Workgroup_Size'])}"
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    ds[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[k]["_v"] = float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"]); agg[k]["_l"] = float(r["LDS_Block_Size"])
lines += ["", "SQ counters (same definitions as the f32 table in `%s_sq_counters.md`):" % tag, "",
          "| kernel | n | avg us | VGPR | LDS B | MfmaBusy | wait | issue-stall | active | LDSconf |",
          "|---|---|---|---|---|---|---|---|---|---|"]
tab = []
for k, v in agg.items():
    n = len(ds[k]); tot = sum(ds[k].values()) / 1e3
    if tot < 300: continue
    wc = max(v["SQ_WAVE_CYCLES"], 1)
    tab.append((tot, f"| `{k}` | {n} | {tot / n:.1f} | {v['_v']:.0f} | {v['_l']:.0f} | "
                f"{v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(4 * v['SQ_BUSY_CU_CYCLES'], 1):.2f} | {v['SQ_WAIT_ANY'] / wc:.2f} | "
                f"{v['SQ_WAIT_INST_ANY'] / wc:.2f} | {v['SQ_ACTIVE_INST_ANY'] / wc:.2f} | "
                f"{v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):.3f} |"))
lines += [t for _, t in sorted(tab, reverse=True)[:16]]
shutil.copy(newest("b16_kt/*/*_kernel_stats.csv"), os.path.join(OUT, f"{tag}_bf16_kernel_stats.csv"))
open(os.path.join(OUT, f"{tag}_bf16_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
