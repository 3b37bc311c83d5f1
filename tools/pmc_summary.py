"""Aggregate rocprofv3 --pmc counter_collection CSVs by kernel (development tool).
usage: python tools/pmc_summary.py <counter_collection.csv> [...more csvs] > summary.json"""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("rau::", "")
    return n[:90]


out = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"]) + " grid=" + r["Grid_Size"]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        out[k]["_ns_" + r["Counter_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[k].add((path, r["Dispatch_Id"]))
        out[k]["VGPR"] = float(r["VGPR_Count"])
        out[k]["LDS"] = float(r["LDS_Block_Size"])
res = {}
for k, v in out.items():
    n = len(cnt[k])
    e = {"dispatches": n, "VGPR": v["VGPR"], "LDS": v["LDS"]}
    for c, x in v.items():
        if c.startswith("_ns_") or c in ("VGPR", "LDS"):
            continue
        e[c] = x / n
    ns = [x for c, x in v.items() if c.startswith("_ns_")]
    e["avg_us"] = (sum(ns) / len(ns)) / n / 1e3 if ns else None
    res[k] = e
json.dump(res, sys.stdout, indent=1, sort_keys=True)
