#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: per kernel, mean of each counter over its dispatches.
usage: tools/pmc_summary.py gpurun_out/pmc_<tag>_*     (also writes profiles/pmc.json: the means, machine-readable)"""
import csv
import json
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if not ("k_shadow" in k or "k_nearest" in k):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for k in acc:
    for fam in ("k_shadow<false>", "k_nearest<false>"):
        if fam in k:
            out[fam.split("<")[0]] = {c: sum(v) / len(v) for c, v in acc[k].items()}
if out:
    json.dump({"method": "rocprofv3 --pmc, means over the launches of one bench run", **out}, open(os.path.join(ROOT, "profiles", "pmc.json"), "w"), indent=1)
