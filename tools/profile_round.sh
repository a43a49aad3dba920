#!/bin/bash
# One round of measurements on the GPU box: counters (SQ, HBM traffic) of the three bench workloads, the bench line,
# kernel traces.  usage: tools/profile_round.sh <tag> [notests]   (outputs under gpurun_out/; copy the summaries to profiles/)
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1 || { tail -20 gpurun_out/${TAG}_tests.log; exit 1; }
  tail -2 gpurun_out/${TAG}_tests.log
fi
# counters first: bench.py replays them only if they were collected on this tree's kernel sources
counters() {  # scene w h suffix
  local S=$1 W=$2 H=$3 SUF=$4
  bash tools/traffic.sh $S $W $H traffic$SUF.json > gpurun_out/${TAG}_traffic$SUF.log 2>&1 || { tail -5 gpurun_out/${TAG}_traffic$SUF.log; return 1; }
  cp profiles/traffic$SUF.json gpurun_out/traffic$SUF.json
  bash tools/pmc.sh raytracer_amd/librtgpu.so ${TAG}$SUF $S $W $H || return 1
  python3 tools/pmc_summary.py --workload "$S ${W}x${H}" --out pmc$SUF.json gpurun_out/pmc_${TAG}${SUF}_1 gpurun_out/pmc_${TAG}${SUF}_2 gpurun_out/pmc_${TAG}${SUF}_3 > gpurun_out/${TAG}${SUF}_pmc_sq_counters.txt
  cp profiles/pmc$SUF.json gpurun_out/pmc$SUF.json
}
counters temple 3840 2160 "" || exit 1
counters ulica 1920 1080 _ulica1080 || exit 1
counters ulica 7680 4320 _ulica8k || exit 1
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
# kernel traces, one workload per run (the statistics are per kernel name), one frame in flight (a kernel's time is its own:
# what the bench line's roofline object is priced with)
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-config5 --frames-in-flight 1 > $ROOT/gpurun_out/${TAG}_trace.log 2>&1) || { tail -5 gpurun_out/${TAG}_trace.log; exit 1; }
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_trace_ulica1080 -- python3 $ROOT/bench.py --no-cpu-baseline --scene ulica --width 1920 --height 1080 --steps 3 --warmup 1 --frames-in-flight 1 > $ROOT/gpurun_out/${TAG}_trace_ulica1080.log 2>&1) || { tail -5 gpurun_out/${TAG}_trace_ulica1080.log; exit 1; }
# (large per-dispatch tables are not needed back: keep the statistics)
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
echo profile round $TAG done
