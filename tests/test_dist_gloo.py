"""The multi-GPU exchange (min/max all-reduce, tile gather, de-interleave) on CPU tensors with the gloo
backend, world_size 2 and 3: every rank takes its round-robin tiles of an oracle-rendered frame, and
rank 0 must reassemble exactly the frame and the global colour range."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as D
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, rgb, ids, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ntiles = D.tiles_of(w, h)
        mine_rgb = D.split_tiles_reference(rgb, rank, world)           # [tiles, 64, 3]
        mine_ids = D.split_tiles_reference(ids, rank, world)           # [tiles, 64]
        assert mine_ids.shape[0] == D.tile_count(ntiles, rank, world)
        inside = D.split_tiles_reference(np.ones((h, w), bool), rank, world)
        px = mine_rgb[inside]
        local = np.concatenate([px.min(0), px.max(0)]).astype(np.float32) if px.size else np.array([3.4e38] * 3 + [1.17e-38] * 3, np.float32)
        mm = D.exchange_minmax(torch.from_numpy(local))
        # planar lights: every rank ends up with every rank's per-pixel counts, each rank's block per_rank long
        counts = ((mine_ids >= 0) & inside).reshape(-1).astype(np.int32)  # (tiles are padded with zeros outside the image)
        every, per_rank, total = D.all_gather_counts(torch.from_numpy(counts), ntiles)
        assert per_rank == D.tile_count(ntiles, 0, world) * 64 and every.numel() == world * per_rank
        assert total == int((ids >= 0).sum())
        for r in range(world):
            want = ((D.split_tiles_reference(ids, r, world) >= 0) & D.split_tiles_reference(np.ones((h, w), bool), r, world)).reshape(-1).astype(np.int32)
            got = every[r * per_rank:(r + 1) * per_rank].numpy()
            assert np.array_equal(got[:want.size], want) and not got[want.size:].any()
        g_ids = D.gather_tiles(torch.from_numpy(mine_ids.reshape(-1).copy()), ntiles, 1)
        g_rgb = D.gather_tiles(torch.from_numpy(mine_rgb.reshape(-1).copy()), ntiles, 3)
        if rank == 0:
            out_q.put((mm.numpy(), D.untile_reference(g_ids.numpy(), world, w, h),
                       D.untile_reference(g_rgb.numpy().reshape(-1, 3), world, w, h)))
        else:
            assert g_ids is None and g_rgb is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_exchange_reassembles_frame(world):
    w, h = 61, 45  # not multiples of 8; tile count not a multiple of world
    o = O.OracleScene(os.path.join(ROOT, "scenes/s3/s3"), width=w, height=h).render(shadow_cache=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, o["rgb"], o["ids"], q)) for r in range(world)]
    for p in procs:
        p.start()
    mm, ids, rgb = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(ids, o["ids"])
    assert np.array_equal(rgb.view(np.uint32), o["rgb"].view(np.uint32))
    flat = o["rgb"].reshape(-1, 3)
    assert np.array_equal(mm, np.concatenate([flat.min(0), flat.max(0)]))
