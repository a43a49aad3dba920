"""raytracer_amd — Python bindings over the two native libraries of this package:

* ``librthost.so``: the C host layer (scene/.cfg loaders, BMP writer, reference-compatible entry points),
* ``librtgpu.so``:  the HIP implementation of the per-pixel trace path behind the C ABI of ``include/rtgpu.h``.

Python is plumbing here (ctypes calls, numpy/torch buffers, torch.distributed for the multi-GPU
exchange); all rendering happens in the HIP kernels.  Importing works without a GPU; creating a
``Renderer`` does not: there is no CPU rendering path.
"""
from .api import (Camera, Counters, csrc_hash, GridInfo, PixelLayout, Renderer, RtgpuError, Scene, Timings, build_native, host_lib,
                  gpu_lib, render_bmp, save_bmp24, scene_paths, TILE, TILE_PIXELS)

__all__ = ["Camera", "Counters", "GridInfo", "PixelLayout", "Renderer", "RtgpuError", "Scene", "Timings", "build_native",
           "host_lib", "gpu_lib", "csrc_hash", "render_bmp", "save_bmp24", "scene_paths", "TILE", "TILE_PIXELS"]
