"""The frame loop of one rank with its collectives really issued to RCCL (a group of one rank, so that it runs on a
one-GPU box): RCCL must accept exactly the tensors the multi-GPU path hands it (split views as gather / all-gather
lists, the 6-float range), and the frame must equal the plain single-GPU frame."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
import raytracer_amd as rt
from raytracer_amd import dist as D
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
D.FORCE_COLLECTIVES = True
for prefix, w, h in (("scenes/s3/s3", 96, 72), ("scenes/s2/s2", 64, 48), ("scenes/s3/s3", 61, 45)):
    scene = rt.Scene(prefix, width=w, height=h)
    r = rt.Renderer(0)
    fr = D.FrameRenderer(r, scene, 0, 1)
    for _ in range(2):
        img = fr.frame()
    torch.cuda.synchronize()
    got = img.cpu().numpy().view(np.uint32).reshape(h, w)
    r2 = rt.Renderer(0)
    r2.upload(scene); r2.frame_begin()
    g = r2.render_host()
    want = r2.tonemap_host(g["rgb"], g["minmax"], scene.gamma)
    assert np.array_equal(got, want), prefix
    assert np.array_equal(fr.global_minmax.cpu().numpy(), g["minmax"]), prefix
    # the same frames two and three deep (what bench.py runs on up to four, and on eight GPUs): each slot of the pipeline has
    # its own context and stream, the collectives of one frame are issued between the kernels of the others
    for depth in (2, 3):
        pipe = D.FramePipeline(lambda: rt.Renderer(0), scene, 0, 1, depth=depth)
        imgs = [pipe.frame() for _ in range(2 * depth + 1)]
        pipe.finish()
        torch.cuda.synchronize()
        for img, _ in imgs[-depth:]:
            assert np.array_equal(img.cpu().numpy().view(np.uint32).reshape(h, w), want), (prefix, depth)
        pipe.close()
    r.close(); r2.close()
dist.destroy_process_group()
print("rccl single rank ok")
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("queues", ["", "16"])
def test_frame_loop_with_collectives_issued_to_rccl(queues):
    """queues: GPU_MAX_HW_QUEUES as the box has it, and 16 as bench.py sets it for its ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if queues:
        env["GPU_MAX_HW_QUEUES"] = queues
    p = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "rccl single rank ok" in p.stdout, (p.stdout[-800:], p.stderr[-1500:])
