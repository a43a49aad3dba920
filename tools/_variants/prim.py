import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, raytracer_amd as rt
from tools.diag import SCENES
scene = rt.Scene(SCENES["temple"], width=3840, height=2160)
for pipe in (1, 0):
    r = rt.Renderer(0, timing=1)
    r.set_option("pipeline", pipe)
    r.upload(scene)
    dev = torch.device("cuda:0")
    r.set_stream(torch.cuda.current_stream(dev))
    for wd in (1, 8, 64):
        r.frame_begin(); n = r.tile_count(0, wd) * 64
        rgb = torch.empty(n * 3, dtype=torch.float32, device=dev); ids = torch.empty(n, dtype=torch.int32, device=dev)
        mm = torch.empty(6, dtype=torch.float32, device=dev)
        for rep in range(3):
            r.timings()
            for _ in range(10):
                r.frame_begin(); r.render(rgb, ids, mm, 0, wd)
            torch.cuda.synchronize(dev)
        t = {k: round(v / 10, 3) for k, v in r.timings().asdict().items() if k.endswith("_ms")}
        print("pipeline", pipe, "world", wd, t, flush=True)
    r.close()
