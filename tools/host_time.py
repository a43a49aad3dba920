#!/usr/bin/env python3
"""Host time per frame of the multi-GPU frame loop, rehearsed on one GPU: rank 0's share of `world` ranks with the
collectives issued to RCCL in a group of one rank.  Prints the time the host needs to enqueue a frame (no GPU wait)
next to the time per frame with the GPU.  usage: tools/host_time.py [world] [depth]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import dist as D  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
D.FORCE_COLLECTIVES = True
scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=3840, height=2160)
pipe = D.FramePipeline(lambda: rt.Renderer(0), scene, 0, world, depth)
for _ in range(4):
    pipe.frame()
torch.cuda.synchronize(dev)
K = 40
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        pipe.frame()
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    print("share of %d ranks, %d in flight: host enqueues a frame in %.3f ms, %.3f ms per frame with the GPU" %
          (world, depth, (t1 - t0) * 1e3 / K, (t2 - t0) * 1e3 / K), flush=True)
pipe.close()
dist.destroy_process_group()
