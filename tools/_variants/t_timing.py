import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import raytracer_amd as rt
scene = rt.Scene("/root/repo/scenes/temple/temple", width=3840, height=2160)
dev = torch.device("cuda:0")
for timing in (1, 0, 1, 0):
    r = rt.Renderer(0, timing=timing)
    r.upload(scene); r.set_stream(torch.cuda.current_stream(dev)); r.frame_begin()
    n = r.tile_count(0, 1) * 64
    rgb = torch.empty(n * 3, dtype=torch.float32, device=dev); ids = torch.empty(n, dtype=torch.int32, device=dev)
    mm = torch.empty(6, dtype=torch.float32, device=dev); rgba = torch.empty(n, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(4):
        r.timings(); torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(10):
            r.frame_begin(); r.render(rgb, ids, mm, 0, 1); r.tonemap(rgb, n, mm, scene.gamma, rgba)
        torch.cuda.synchronize(dev); best = min(best, (time.perf_counter() - t0) * 100)
    print("timing=%d: %.3f ms per frame" % (timing, best), flush=True)
    r.close()
