#!/bin/bash
# HBM traffic of the walk kernels from the L2 memory-side counters, one counter per rocprofv3 run.
# usage: tools/traffic.sh [scene w h [out.json]]   -> gpurun_out/traffic_*/ and profiles/<out.json> (default traffic.json)
SCENE=${1:-temple}; W=${2:-3840}; H=${3:-2160}; OUT=${4:-traffic.json}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for P in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $P | tr ' ' '_')
  rm -rf $ROOT/gpurun_out/traffic_$T
  timeout -s KILL 180 rocprofv3 --pmc $P --output-format csv -d $ROOT/gpurun_out/traffic_$T -- python3 $ROOT/tools/kbench.py --one $SCENE $W $H > $ROOT/gpurun_out/traffic_$T.log 2>&1 || { echo "pass $P failed"; tail -3 $ROOT/gpurun_out/traffic_$T.log; exit 1; }
done
python3 $ROOT/tools/traffic_summary.py $ROOT/gpurun_out $SCENE $W $H $OUT
