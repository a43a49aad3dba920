#!/bin/bash
# Builds the walk kernels with the phase clocks in (-DRT_PHASES) into tools/_variants/ (run here, where hipcc is
# cheap), then `python tools/phases.py [scene w h]` on the GPU box prints each phase's share of the wave time.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tools/_variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -DRT_PHASES=${RT_PHASES:-1} \
  -shared -o $ROOT/tools/_variants/librtgpu_phases.so $ROOT/raytracer_amd/csrc/rtgpu.hip
