#!/usr/bin/env python3
"""Benchmark of the per-pixel trace path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene temple] [--width 3840 --height 2160]

Workload (N=1 and N>1 alike): scenes/temple at 3840x2160, the configuration the headline metric is
quoted on.  One step = one frame = everything the reference's rtVisualizedSceneRaytrace +
rtVisualizedSceneToBitmap do: per-triangle preprocess, grid build, primary rays, nearest-hit walks,
shading with one shadow query per (hit, light), global min/max, tone map to RGBA8, and with N>1 the
min/max all-reduce and the RGBA8 tile gather to rank 0 (RCCL).  The scene is uploaded to HBM before
the timed region; the finished image stays in HBM on rank 0.  The frame is split by 8x8 tiles over
the ranks, so total work is fixed as N grows ("strong" scaling).

metric = Mrays/s, rays counted as the reference issues them (nearest-hit queries + shadow queries,
back-face early-outs included): 40,689,192 per temple 4K frame.

Extra objects on the JSON line:
  roofline      the dominant kernel (shadow or nearest walk): algorithmic bytes per launch
                (64 B per ray + 8 B per visited voxel + 40 B per listed triangle, S and T counted by the
                kernel's own counting variant in an untimed pass) / its average launch duration from HIP
                events on the launch stream.  The scene is cache-resident and most listed triangles are
                rejected a batch at a time, so this logical rate exceeds the HBM peak by a wide margin;
                `traffic` is what the launch really moved to and from HBM (rocprofv3 counters), and
                walk_stats says how much of the listed work was executed.
  cpu_baseline  the unmodified reference (oracle/_ref/ref_dump) or, if that binary is absent, the C
                oracle, single-threaded on the host, on the same scene at a reduced resolution;
                cpu_baseline.parallel is the C oracle on the same sample with its rows spread over the
                host cores of this job (OpenMP).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # HBM3E peak, GB/s


def cpu_baseline(prefix, over, width, height):
    """Times the CPU path on one host core on a bounded sample: the same scene at width x height."""
    import raytracer_amd as rt
    from oracle import oracle as O
    # rays of the sample, counted by the reference's own definition (the GPU path issues the same queries)
    scene = rt.Scene(prefix, width=width, height=height, **over)
    r = rt.Renderer(0)
    r.upload(scene)
    r.frame_begin()
    r.render_host()
    rays = r.counters().rays
    r.close()
    sample = "%s at %dx%d, whole frame, 1 thread" % (os.path.basename(prefix), width, height)
    if O.have_reference():
        ref = O.run_reference(prefix, width=width, height=height, cwd=os.path.join(ROOT, "scenes"), **over)
        secs, kind = ref["seconds"], "reference"
    else:
        s = O.OracleScene(prefix, width=width, height=height, **over)
        t0 = time.time()
        s.render(shadow_cache=True, threads=1)
        secs, kind = time.time() - t0, "port"
    out = {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": kind, "sample": sample,
           "seconds": round(secs, 3), "rays": rays}
    # The same sample with the C oracle's rows spread over the host cores this job may use (the reference itself has no
    # threads; rows in parallel need its per-light shadow cache off, so more shadow queries reach the grid walk).
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), 64))
    if cores > 1 and scene.npl == 0:
        s = O.OracleScene(prefix, width=width, height=height, **over)
        t0 = time.time()
        s.render(shadow_cache=False, threads=cores)
        psecs = time.time() - t0
        out["parallel"] = {"value": round(rays / psecs / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                           "sample": sample.replace("1 thread", "%d threads, rows in parallel, shadow cache off" % cores),
                           "seconds": round(psecs, 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="temple")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-width", type=int, default=960)
    ap.add_argument("--cpu-height", type=int, default=540)
    args = ap.parse_args()

    # Libraries below write banners to the C-level stdout (RCCL prints its version when a communicator is made); the
    # one JSON line is the only thing this program may print there, so stdout points to stderr until that line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.dist import FrameRenderer

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force = world == 1 and os.environ.get("RTGPU_BENCH_FORCE_COLLECTIVES") == "1"  # rehearsal of the N > 1 code on one GPU
    if world > 1 or force:
        if force:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            import raytracer_amd.dist as rdist
            rdist.FORCE_COLLECTIVES = True
        dist.init_process_group("nccl", device_id=dev)

    scenes = {"temple": ("scenes/temple/temple", {}), "ulica": ("scenes/ulica/ulica", {}), "s3": ("scenes/s3/s3", {}),
              "s5": ("scenes/s5/s5", {}), "pokoj": ("scenes/pokoj/s2", {}), "s2": ("scenes/s2/s2", {})}
    prefix, over = scenes[args.scene]
    prefix = os.path.join(ROOT, prefix)
    scene = rt.Scene(prefix, width=args.width, height=args.height, **over)
    renderer = rt.Renderer(local_rank)

    # untimed counting pass: rays per frame, S and T per kernel family for the roofline (this rank's tiles)
    renderer.set_option("count", 1)
    fr = FrameRenderer(renderer, scene, rank, world)
    fr.frame()
    torch.cuda.synchronize(dev)
    cnt = renderer.counters().asdict()
    renderer.set_option("count", 0)
    renderer.set_option("timing", 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        fr.frame()
    barrier()
    renderer.timings()  # drop what warm-up accumulated
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fr.frame()
    barrier()
    elapsed = time.perf_counter() - t0
    tim = renderer.timings().asdict()

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rays_t = torch.tensor([cnt["nearest_queries"] + cnt["shadow_queries"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_frame = float(rays_t.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        # dominant kernel on this rank
        fam = "shadow" if tim["shadow_ms"] >= tim["nearest_ms"] else "nearest"
        launches = max(tim[fam + "_launches"], 1)
        avg_ms = tim[fam + "_ms"] / launches
        if fam == "shadow":
            q, S, T = cnt["shadow_queries"], cnt["shadow_steps"], cnt["shadow_tests"]
        else:
            q, S, T = cnt["nearest_queries"], cnt["nearest_steps"], cnt["nearest_tests"]
        per_launch = args.steps / launches  # frames per launch of this family (1 when the frame is one chunk)
        alg_bytes = (64.0 * q + 8.0 * S + 40.0 * T) * per_launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, l2_hit = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and world == 1:  # (the counters were collected on whole-frame launches)
            try:
                tj = json.load(open(tfile)).get("k_" + fam, {})
                traffic = tj.get("hbm_bytes_per_launch")
                l2_hit = tj.get("l2_hit_rate")
            except Exception:
                traffic = None
        # what really bounds the kernel: instruction issue.  Wave-instructions and cycles of one launch from the SQ
        # counters of the last profiled run (profiles/pmc.json); the machine's rate from tools/micro/issue_rate.hip.
        issue = None
        pfile = os.path.join(ROOT, "profiles", "pmc.json")
        if os.path.exists(pfile) and world == 1:
            try:
                pm = json.load(open(pfile)).get("k_" + fam, {})
                insts = sum(pm.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"))
                cycles = pm.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
                if insts > 0 and cycles > 0:
                    cpi = cycles * 1024.0 / insts  # 256 CUs x 4 SIMDs
                    waves = pm.get("SQ_WAVES", 0.0)
                    issue = {"wave_instructions_per_launch": int(insts), "cycles_per_launch": int(cycles),
                             "valu_share_of_instructions": round(pm.get("SQ_INSTS_VALU", 0.0) / insts, 3),
                             # a wave64 VALU instruction occupies the 32-lane SIMD for two cycles
                             "valu_busy_frac": round(2.0 * pm.get("SQ_INSTS_VALU", 0.0) / (cycles * 1024.0), 3),
                             "resident_waves_per_simd": round(waves / 1024.0, 2) if waves else None,
                             "cycles_per_instruction_per_simd": round(cpi, 3), "machine_cycles_per_instruction": 2.1,
                             "frac": round(2.1 / cpi, 3), "source": "profiles/pmc.json (rocprofv3 --pmc of an earlier run of this workload)"}
            except Exception:
                issue = None
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": round(rays_frame * args.steps / elapsed / 1e6, 3),
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "bundled scene scenes/%s (the reference's own input), camera resolution set to %dx%d"
                                    % (args.scene, args.width, args.height),
            "config": {"workload": "scenes/%s %dx%d, depth 5, %d point lights, frame = preprocess + grid build + trace + "
                                   "shade + tone map%s" % (args.scene, args.width, args.height, scene.nl,
                                                            " + RCCL min/max all-reduce + RGBA8 tile gather" if world > 1 else ""),
                       "rays_per_frame": int(rays_frame), "tiling": "8x8 tiles round-robin over %d rank(s)" % world},
            "kernel_ms_per_frame_rank0": {k: round(v / args.steps, 3) for k, v in tim.items() if k.endswith("_ms")},
            "kernel_ms_note": "HIP events on the stream each kernel runs on; preprocess_ms and grid_ms are elapsed on the build "
                              "stream, where the set-up of frame k+1 overlaps the walk kernels of frame k (alone it takes 0.27 ms)",
            "roofline": {"bound": "hbm", "kernel": "k_" + fam, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "hbm_gbs_measured": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic and avg_ms > 0 else None,
                         "l2_hit_rate": round(l2_hit, 4) if l2_hit is not None else None,
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                         "listed_tests_per_launch": int(T * per_launch),
                         "executed_tests_frac": round(cnt[fam + "_lane_slots"] / T, 4) if T else None,
                         "note": "logical bytes (64 B/ray + 8 B/voxel + 40 B/listed triangle); the scene is cache-resident and "
                                 "batches of listed triangles are culled, so frac exceeds 1 without touching HBM; the kernel is "
                                 "bound by instruction issue (see profiles/)", "issue": issue},
        }
        def ratio(a, b):
            return round(a / b, 4) if b else None
        out["walk_stats_rank0"] = {
            f: {"simd_efficiency": ratio(cnt[f + "_lane_tests"], cnt[f + "_lane_slots"]),
                "wave_tests": cnt[f + "_lane_slots"] // 64, "wave_cell_visits": cnt[f + "_visits"],
                "batches_culled": ratio(cnt[f + "_batches_culled"], cnt[f + "_batches"]),
                "reached_second_slabs": ratio(cnt[f + "_stage2"], cnt[f + "_lane_slots"] / 64),
                "reached_distance_stage": ratio(cnt[f + "_stage3"], cnt[f + "_lane_slots"] / 64),
                "reached_exact_test": ratio(cnt[f + "_exact"], cnt[f + "_lane_slots"] / 64)}
            for f in ("nearest", "shadow")}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(prefix, over, args.cpu_width, args.cpu_height)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": str(e)}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
