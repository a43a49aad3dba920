#!/bin/bash
# One round of measurements on the GPU box: tests, bench, kernel trace, SQ counters, HBM traffic.
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/, summaries copied to profiles/ afterwards)
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1 || { tail -20 gpurun_out/${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests.log
# counters first: bench.py replays them only if they were collected on this tree's kernel sources
bash tools/traffic.sh > gpurun_out/${TAG}_traffic.log 2>&1 || { tail -5 gpurun_out/${TAG}_traffic.log; exit 1; }
cp profiles/traffic.json gpurun_out/traffic.json
bash tools/pmc.sh raytracer_amd/librtgpu.so $TAG || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 > gpurun_out/${TAG}_pmc_sq_counters.txt
cp profiles/pmc.json gpurun_out/pmc.json
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_trace -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_trace.log 2>&1) || { tail -5 gpurun_out/${TAG}_trace.log; exit 1; }
echo profile round $TAG done

