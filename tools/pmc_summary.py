#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: per kernel, mean of each counter over its dispatches.
usage: tools/pmc_summary.py [--workload "temple 3840x2160"] [--out pmc.json] gpurun_out/pmc_<tag>_*
Also writes profiles/<out> (default pmc.json): the means, machine-readable, stamped with the workload and with the SHA-256
of the kernel sources they were collected on (bench.py picks the file by its workload and ignores it when the sources
differ from its own tree's)."""
import csv
import json
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.api import csrc_hash  # noqa: E402

argv = sys.argv[1:]
workload = "temple 3840x2160"
out_name = "pmc.json"
while argv and argv[0] in ("--workload", "--out"):
    if argv[0] == "--workload":
        workload = argv[1]
    else:
        out_name = argv[1]
    argv = argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in argv:
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

out = {}
for k in acc:
    for fam in ("k_shadow<false", "k_nearest<false"):
        if fam in k:
            out[fam.split("<")[0]] = {c: sum(v) / len(v) for c, v in acc[k].items()}
for k in acc:
    # the shadow queries of a frame are two launches where the light lists have a kernel of their own (k_shadow_lists, then
    # k_shadow for what it leaves): one unit, the counters of the two added (SQ_WAVES is then the waves of both)
    if "k_shadow_lists" in k and "k_shadow" in out:
        for c, v in acc[k].items():
            out["k_shadow"][c] = out["k_shadow"].get(c, 0.0) + sum(v) / len(v)
        out["k_shadow"]["launches_added"] = 2
if out:
    json.dump({"method": "rocprofv3 --pmc, means over the launches of one run of tools/kbench.py --one", "workload": workload,
               "csrc_sha256": csrc_hash(), **out}, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
