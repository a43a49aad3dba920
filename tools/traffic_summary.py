#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes of tools/traffic.sh into profiles/traffic.json.

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte fabric reads at 64 bytes, so
read bytes = 2 x FETCH_SIZE x 1024 (exact for wide coalesced streams; an upper-bound style correction for
the scattered reads of these kernels); WRITE_SIZE is exact for 16-byte stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, scene, w, h = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
out_name = sys.argv[5] if len(sys.argv) > 5 else "traffic.json"  # file under profiles/ (bench.py picks it by its workload)
sys.path.insert(0, os.path.dirname(os.path.abspath(root)))
from raytracer_amd.api import csrc_hash  # noqa: E402
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "traffic_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "").split("<")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"workload": "%s %sx%s" % (scene, w, h), "csrc_sha256": csrc_hash(), "method": "rocprofv3 --pmc, one counter set per run; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"}
for k in ("k_nearest", "k_shadow", "k_hit", "k_accumulate"):
    if k not in acc:
        continue
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    if k == "k_shadow" and "k_shadow_lists" in acc:
        # the shadow queries of a frame are two launches there (k_shadow_lists, then k_shadow for what it leaves): one unit
        for c, v in acc["k_shadow_lists"].items():
            m[c] = m.get(c, 0.0) + sum(v) / len(v)
    fetch, write = m.get("FETCH_SIZE", 0.0), m.get("WRITE_SIZE", 0.0)
    hit, miss = m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0)
    out[k] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
              "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "l2_requests_per_launch": hit + miss, "launches_averaged": len(next(iter(acc[k].values())))}
json.dump(out, open(os.path.join(os.path.dirname(root), "profiles", out_name), "w"), indent=1)
print(json.dumps(out, indent=1))
