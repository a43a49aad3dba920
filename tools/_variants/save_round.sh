#!/bin/bash
# Copies the summaries of a profile round (tools/_variants/run_final.sh <tag>, merged back under gpurun_out/) into profiles/.
set -e
T=$1; G=gpurun_out
cp $G/pmc.json /tmp/pmc_keep.json
cp $G/traffic.json profiles/traffic.json; cp $G/traffic.json profiles/${T}_traffic.json
cp $G/${T}_bench.json profiles/${T}_bench_temple4k.json; cp $G/${T}_pmc_sq_counters.txt profiles/${T}_pmc_sq_counters.txt
cp $(ls $G/${T}_trace/vm/*kernel_stats.csv) profiles/${T}_kernel_stats.csv
cp $(ls $G/${T}_ulica_trace/vm/*kernel_stats.csv) profiles/${T}_ulica1080_kernel_stats.csv
cp $G/${T}_ulica1080_bench.json profiles/${T}_ulica1080_bench.json
python3 tools/pmc_summary.py $G/pmc_${T}_ulica_1 $G/pmc_${T}_ulica_2 $G/pmc_${T}_ulica_3 > profiles/${T}_ulica1080_pmc_sq_counters.txt
cp /tmp/pmc_keep.json profiles/pmc.json; cp /tmp/pmc_keep.json $G/pmc.json
python3 - <<'PY'
import json
from raytracer_amd import csrc_hash
h = csrc_hash()
for f in ['profiles/pmc.json', 'profiles/traffic.json']:
    print(f, "matches the tree:", json.load(open(f)).get('csrc_sha256') == h)
PY
cat $G/${T}_share8.log $G/${T}_ulica8k.log
