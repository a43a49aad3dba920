#!/usr/bin/env python3
"""The most expensive tickets (64 consecutive rays) of the two walk kernels for one rank's share of a frame.
Needs a library built with -DRT_TICKETS:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -DRT_TICKETS -DRT_PHASES=2 \
        -shared -o tools/_variants/librtgpu_tickets.so raytracer_amd/csrc/rtgpu.hip
usage: python tools/tickets.py [world [scene w h [waves_per_cu]]]     (waves_per_cu 4 = every wavefront alone on its SIMD)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RTGPU_LIB", os.path.join(ROOT, "tools/_variants/librtgpu_tickets.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from tools.diag import SCENES  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    name = sys.argv[2] if len(sys.argv) > 2 else "temple"
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
    scene = rt.Scene(os.path.join(ROOT, SCENES[name]), width=w, height=h)
    r = rt.Renderer(0, timing=1)
    if len(sys.argv) > 5:
        r.set_option("waves_per_cu", int(sys.argv[5]))
    r.upload(scene)
    dev = torch.device("cuda:0")
    r.set_stream(torch.cuda.current_stream(dev))
    r.frame_begin()
    n = r.tile_count(0, world) * 64
    rgb = torch.empty(n * 3, dtype=torch.float32, device=dev)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    mm = torch.empty(6, dtype=torch.float32, device=dev)
    for _ in range(2):
        r.frame_begin()
        r.timings()
        r.render(rgb, ids, mm, 0, world)
    t = r.timings().asdict()
    L = rt.gpu_lib()
    # how much of their life the wavefronts spend inside tickets, and how much longer the last one lives than the average
    L.rtgpu_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    life = (C.c_int64 * 16)()
    L.rtgpu_phase_cycles(r._h, life, 1)  # (sums over the frames rendered above: the ratios stay what they are)
    for which, kind in enumerate(("nearest", "shadow")):
        busy, alive, waves, longest = life[8 * which:8 * which + 4]
        if waves:
            print("%s walk: wavefronts spend %.1f %% of their life inside tickets; mean life %.0f clocks, longest %d"
                  % (kind, 100.0 * busy / alive, alive / waves, longest))
    L.rtgpu_ticket_log.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    tiles_x = (w + 7) // 8
    for which, kind in enumerate(("nearest", "shadow")):
        buf = np.zeros((1 << 18, 12), np.uint32)
        got = C.c_int(0)
        rc = L.rtgpu_ticket_log(r._h, which, buf.ctypes.data, buf.shape[0], C.byref(got))
        assert rc == 0 and got.value > 0, "library built without -DRT_TICKETS?"
        used = buf[:, 0] > 0
        log = buf[used].astype(np.int64)
        idx = np.nonzero(used)[0]
        cyc = log[:, 0]
        print("%s walk %.3f ms: %d tickets, clocks mean %.0f, median %.0f, p99 %.0f, max %d (%.1f x mean); sum/8192 waves = %.0f"
              % (kind, t[kind + "_ms"], len(log), cyc.mean(), np.median(cyc), np.quantile(cyc, 0.99), cyc.max(), cyc.max() / cyc.mean(), cyc.sum() / 8192))
        print("   per ticket: visits %.1f, batches (view / light lists: references read) %.1f, records tested %.1f" % (log[:, 1].mean(), log[:, 2].mean(), log[:, 3].mean()))
        if log[:, 4:].sum() > 0:
            ph = log[:, 4:].sum(0)
            print("   all tickets, share of the phases (set-up, mode, headers+cos, culling, record staging, tests, stepping, lane/ray mode): "
                  + " ".join("%.1f%%" % (100.0 * x / ph.sum()) for x in ph))
        for j in np.argsort(-cyc)[:12]:
            where = ""
            if which == 0:
                tile = idx[j] * world
                where = " tile (%d, %d)" % (tile % tiles_x * 8, tile // tiles_x * 8)
            print("   ticket %7d%s: clocks %8d  visits %4d  batches tested ray by ray %5d  records tested %6d" % (idx[j], where, *log[j][:4]))
            if log[j][4:].sum() > 0:
                print("        phases (set-up, mode, headers+cos, culling, record staging, tests, stepping, lane/ray mode): " + " ".join("%d" % x for x in log[j][4:]))


if __name__ == "__main__":
    main()
