#!/bin/bash
# usage: tools/pmc.sh <lib.so> <tag> [scene w h]  -- three counter passes of one frame loop under rocprofv3
# (each pass its own run; outputs gpurun_out/pmc_<tag>_<pass>/)
LIB=$(realpath "$1"); TAG=$2; SCENE=${3:-temple}; W=${4:-3840}; H=${5:-2160}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export RTGPU_LIB=$LIB TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  rm -rf $ROOT/gpurun_out/pmc_${TAG}_$i
  timeout -s KILL 180 rocprofv3 --pmc $P --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$i -- python3 $ROOT/tools/kbench.py --one $SCENE $W $H > $ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $ROOT/gpurun_out/pmc_${TAG}_$i.log; exit 1; }
  i=$((i+1))
done
echo done $TAG
