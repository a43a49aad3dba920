#!/usr/bin/env python3
"""Where the time of the drop-in frame call goes: the calls rtVisualizedSceneRaytrace makes (integration/raytrace_gpu.c),
timed one by one on a temple frame, and the rate of the host link into page-locked and pageable memory.
usage: tools/dropin_time.py [scene w h]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402

scene_name = sys.argv[1] if len(sys.argv) > 1 else "temple"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
prefix = {"temple": "scenes/temple/temple", "ulica": "scenes/ulica/ulica", "s3": "scenes/s3/s3"}[scene_name]
scene = rt.Scene(os.path.join(ROOT, prefix), width=w, height=h)
L = rt.gpu_lib()
r = rt.Renderer(0)
lay = rt.PixelLayout(24, 0, 16, 0x10000, 232)
n = w * h * 24
block = L.rtgpu_pinned_alloc(n)
pageable = (C.c_ubyte * n)()
mm = np.zeros(6, np.float32)


def timed(f, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


r.upload(scene); r.frame_begin()
L.rtgpu_render_pixelmap(r._h, C.c_void_p(block), C.byref(lay), mm.ctypes.data_as(C.c_void_p))
print("scene_upload (same scene)   %.3f ms" % timed(lambda: r.upload(scene)))
print("frame_begin (enqueue)       %.3f ms" % timed(lambda: r.frame_begin()))
r.sync()
print("render_pixelmap, pinned     %.3f ms" % timed(lambda: (r.frame_begin(), L.rtgpu_render_pixelmap(r._h, C.c_void_p(block), C.byref(lay), mm.ctypes.data_as(C.c_void_p)))))
print("render_pixelmap, pageable   %.3f ms" % timed(lambda: (r.frame_begin(), L.rtgpu_render_pixelmap(r._h, C.cast(pageable, C.c_void_p), C.byref(lay), mm.ctypes.data_as(C.c_void_p)))))
import torch  # noqa: E402
d = torch.empty(n, dtype=torch.uint8, device="cuda")
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hq = torch.empty(n, dtype=torch.uint8)
torch.cuda.synchronize()
for name, dst in (("pinned", hp), ("pageable", hq)):
    ms = timed(lambda: (dst.copy_(d, non_blocking=True), torch.cuda.synchronize()))
    print("device -> %-9s %d MB   %.3f ms = %.1f GB/s" % (name, n // 1000000, ms, n / ms / 1e6))
ms = timed(lambda: (d.copy_(hp, non_blocking=True), torch.cuda.synchronize()))
print("pinned -> device    %d MB   %.3f ms = %.1f GB/s" % (n // 1000000, ms, n / ms / 1e6))
for off in (64, 256, 4096):
    b2 = L.rtgpu_pinned_alloc(n + off)
    print("render_pixelmap, pinned + %-4d %.3f ms" % (off, timed(lambda: (r.frame_begin(), L.rtgpu_render_pixelmap(r._h, C.c_void_p(b2 + off), C.byref(lay), mm.ctypes.data_as(C.c_void_p))))))
    L.rtgpu_pinned_free(b2)
tri = np.zeros((scene.nt, 10), np.float32)
print("triangles_download          %.3f ms" % timed(lambda: L.rtgpu_triangles_download(r._h, tri.ctypes.data_as(C.c_void_p))))
def seq():
    r.upload(scene); r.frame_begin()
    L.rtgpu_render_pixelmap(r._h, C.c_void_p(block), C.byref(lay), mm.ctypes.data_as(C.c_void_p))
    r.grid_info(); L.rtgpu_triangles_download(r._h, tri.ctypes.data_as(C.c_void_p))
print("the stub's sequence         %.3f ms" % timed(seq))
L.rtgpu_pinned_free(block)
r.close()
# the host layer's own part of the call (raytracer_amd/host/raytrace.c)
H = rt.host_lib()
H.rtVisualizedSceneRaytrace.restype = C.c_void_p
H.rtVisualizedSceneRaytrace.argtypes = [C.c_void_p, C.c_void_p]
H.rtVisualizedSceneDestroy.argtypes = [C.POINTER(C.c_void_p)]
H.rtSceneFlatFree.argtypes = [C.POINTER(C.POINTER(rt.api.SceneFlat))]
cam = C.cast(scene._cam, C.POINTER(rt.Camera))  # RT_Camera: four RT_Vertex4f, then sw, sh
class HostCam(C.Structure):
    _fields_ = [("v", C.c_float * 16), ("sw", C.c_int32), ("sh", C.c_int32)]
hc = C.cast(scene._cam, C.POINTER(HostCam)).contents
hc.sw, hc.sh = w, h
print("rtScenePreprocess           %.3f ms" % timed(lambda: H.rtScenePreprocess(scene._scene, scene._cam)))
def flat():
    f = H.rtSceneFlatten(scene._scene); H.rtSceneFlatFree(C.byref(f))
print("rtSceneFlatten + free       %.3f ms" % timed(flat))
def whole():
    vs = C.c_void_p(H.rtVisualizedSceneRaytrace(scene._scene, scene._cam))
    H.rtVisualizedSceneDestroy(C.byref(vs))
whole()
print("rtVisualizedSceneRaytrace   %.3f ms" % timed(whole))
