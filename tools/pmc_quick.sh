#!/bin/bash
# Instruction counts of the walk kernels for a library: tools/pmc_quick.sh <lib.so> <tag> [scene w h] -> one counter pass, printed
LIB=$(realpath "$1"); TAG=$2; SCENE=${3:-temple}; W=${4:-3840}; H=${5:-2160}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export RTGPU_LIB=$LIB TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pq_${TAG} -- python3 $ROOT/tools/kbench.py --one $SCENE $W $H > $ROOT/gpurun_out/pq_${TAG}.log 2>&1 || { echo "pass failed"; tail -3 $ROOT/gpurun_out/pq_${TAG}.log; exit 1; }
python3 - "$ROOT/gpurun_out/pq_${TAG}" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_shadow" in k or "k_nearest" in k:
            acc[k.split("<")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, " ".join("%s=%.4g" % (c.replace("SQ_INSTS_", ""), sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
