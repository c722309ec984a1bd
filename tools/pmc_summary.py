#!/usr/bin/env python3
"""Summarises the counter CSVs of tools/pmc_rdreq.sh: per kernel, mean counter value per dispatch.
usage: tools/pmc_summary.py <outdir>  -> <outdir>/summary.json"""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_score" in k or "k_ed_" in k or "k_gen" in k or "k_voxel" in k or "k_pairs" in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
res = {}
for (k, c), v in sorted(agg.items()):
    res.setdefault(k, {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k in res:
    print(k)
    for c, v in res[k].items():
        print("   %-40s n=%d mean=%.6g" % (c, v["dispatches"], v["mean"]))
