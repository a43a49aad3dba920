#!/bin/bash
# Register use, spills, scratch and LDS of every kernel of librtgpu, as the compiler reports them.
# usage: tools/kres.sh [extra hipcc flags]   (compiles raytracer_amd/csrc/rtgpu.hip to /dev/null)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC "$@" \
  -Rpass-analysis=kernel-resource-usage -c $ROOT/raytracer_amd/csrc/rtgpu.hip -o /dev/null 2>&1 | grep "remark:" |
  awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /TotalSGPRs:/ {s=$4} /VGPRs Spill/ {vs=$5} /SGPRs Spill/ {ss=$5} /ScratchSize/ {sc=$5} /Occupancy/ {oc=$5} /LDS Size/ {print "vgpr", v, "sgpr", s, "vspill", vs, "sspill", ss, "scratch", sc, "occ", oc, "lds", $6, name}' |
  c++filt | sed -e 's/rt:://g' -e 's/(.*//' | sort -k15
