#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: per kernel, mean of each counter over its dispatches.
usage: tools/pmc_summary.py gpurun_out/pmc_<tag>_*"""
import csv
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
