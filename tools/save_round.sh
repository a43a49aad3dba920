#!/bin/bash
# Copies the summaries of one tools/profile_round.sh run from gpurun_out/ (scratch) to profiles/ (kept).
# usage: tools/save_round.sh <tag>
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
for W in "" _ulica1080 _ulica8k; do
  cp gpurun_out/${TAG}${W}_pmc_sq_counters.txt profiles/${TAG}${W}_pmc_sq_counters.txt
  cp gpurun_out/traffic$W.json profiles/${TAG}${W}_traffic.json
  cp gpurun_out/traffic$W.json profiles/traffic$W.json
  cp gpurun_out/pmc$W.json profiles/pmc$W.json
done
for W in "" _ulica1080; do
  f=$(ls gpurun_out/${TAG}_trace$W/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f profiles/${TAG}${W}_kernel_stats.csv
done
ls -la profiles | grep ${TAG}
