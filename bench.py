#!/usr/bin/env python3
"""Benchmark of the per-pixel trace path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene temple] [--width 3840 --height 2160] [--no-config5]

Workload (N=1 and N>1 alike): scenes/temple at 3840x2160, the configuration the headline metric is
quoted on.  One step = one frame = everything the reference's rtVisualizedSceneRaytrace +
rtVisualizedSceneToBitmap do: per-triangle preprocess, grid build, primary rays, nearest-hit walks,
shading with one shadow query per (hit, light), global min/max, tone map to RGBA8, and with N>1 the
min/max all-reduce and the RGBA8 tile gather to rank 0 (RCCL).  The scene is uploaded to HBM before
the timed region; the finished image stays in HBM on rank 0.  The frame is split by 8x8 tiles over
the ranks, so total work is fixed as N grows ("strong" scaling).  Two frames are in flight on every GPU, three from
eight GPUs on (raytracer_amd.dist.FramePipeline, headline_depth: each in its own context and stream; a frame's exchange
and the latency-bound end of its walk kernels run under the other frames' kernels; the process starts HIP with a hardware
queue for every stream, GPU_MAX_HW_QUEUES=16); all K frames are complete when the timed region ends, and
ms_per_step is the region's length / K.  On one GPU the line also carries one_frame_at_a_time: the same workload with one
frame in flight, where a kernel's launch time is its own (value, ms_per_step, kernel times and roofline of that run).
config.frames_in_flight says how many, --frames-in-flight overrides; frame_latency_ms is one frame alone, start to
finish (barriers included).

metric = Mrays/s, rays counted as the reference issues them (nearest-hit queries + shadow queries,
back-face early-outs included): 40,689,192 per temple 4K frame.

Extra objects on the JSON line:
  config5       the baseline's fifth configuration, scenes/ulica with 64 lights and secondary rays to depth 5, on the same
                ranks with the same exchange: "ulica_1920x1080" (three timed frames) and "ulica_7680x4320" (one timed
                frame: 33 M pixels, 5.5 G rays), each with value, ms_per_step, rays_per_frame, the per-kernel table and
                its own roofline object; and the reference timed on the same scene at 160x90 (cpu_baseline).
  roofline      the dominant kernel (shadow or nearest walk) against the bound that holds it, VALU issue: the
                scene is cache-resident, nothing is a dense contraction, and a SIMD issues one wave64 VALU
                instruction per two cycles.  frac = 2 x SQ_INSTS_VALU / (cycles x 1024 SIMDs) of one launch,
                achieved / peak = VALU wave-instructions per second against SIMDs x clock / 2.  Beside it:
                traffic (HBM bytes per launch from the L2's memory-side counters, corrected as measured on
                gfx950: 2 x FETCH_SIZE + WRITE_SIZE), hbm_frac (traffic / launch time / 8 TB/s), l2_gbs,
                l2_hit_rate, flops (45 x exact tests that ran to their last line, per ray) and logical_gbs
                (64 B per ray + 8 B per visited voxel + 40 B per listed triangle over the launch time: what a
                walk that read every listed triangle from memory would move; most of it is culled and the rest
                is served by LDS and L2, so this is no fraction of any peak).  The launch time is measured in
                this run (HIP events on the launch stream; a frame of ulica launches the kernel once per recursion
                level and chunk: the mean over those launches, like the counters); the counter values come from the
                rocprofv3 --pmc passes kept under profiles/ (pmc*.json, traffic*.json: a profiler cannot run inside the
                timed region), one pair of files per workload, which carry the SHA-256 of raytracer_amd/csrc/* they were
                collected on: if that differs from this tree's the counter-derived fields are null and "stale" is true.
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
HW_QUEUES = 16  # hardware queues HIP may share among this process's streams (GPU_MAX_HW_QUEUES; main() says why)
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


def launch_ranks(args, argv, script=None):
    """`python bench.py --gpus N` without a launcher: starts N fresh ranks (torch.distributed.run, one process per
    GPU) as a child of this process, which has touched neither torch nor the GPU, relays rank 0's JSON line and
    returns the child's exit status."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1], flush=True)
    if child.returncode != 0:
        sys.stderr.write("bench.py: the %d-rank run ended with status %d\n" % (args.gpus, child.returncode))
        return child.returncode
    if not lines:
        sys.stderr.write("bench.py: the %d-rank run printed no result line\n" % args.gpus)
        return 1
    return 0


def counter_profile(fam, workload):
    """Counter values of one launch of k_<fam> for `workload` ("temple 3840x2160") from the rocprofv3 passes kept under
    profiles/ (SQ counters in pmc*.json, L2 memory-side counters in traffic*.json; the file whose "workload" member says
    so), or None for a file that is missing or was collected on other kernel sources than this tree's."""
    import glob
    import raytracer_amd as rt
    have = rt.csrc_hash()
    out = {}
    for key, pattern in (("pmc", "pmc*.json"), ("traffic", "traffic*.json")):
        out[key] = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
            try:
                j = json.load(open(path))
            except Exception:
                continue
            if j.get("workload") == workload and j.get("csrc_sha256") == have and ("k_" + fam) in j:
                out[key] = dict(j["k_" + fam], workload=j.get("workload"), file=os.path.basename(path))
    return out


def roofline(fam, avg_ms, q, S, T, full_tests, lane_slots, workload):
    """The roofline object of the dominant walk kernel (see the module docstring)."""
    secs = avg_ms * 1e-3
    # the counters describe the launches of whole frames of one workload on one GPU (workload None: a rank's share, no counters)
    prof = counter_profile(fam, workload) if workload else {"pmc": None, "traffic": None}
    pm, tr = prof["pmc"], prof["traffic"]
    out = {"bound": "valu_issue", "kernel": "k_" + fam, "achieved": None, "peak": None, "unit": "G wave-instructions/s",
           "frac": None, "traffic": None, "hbm_frac": None, "hbm_gbs": None, "l2_gbs": None, "l2_hit_rate": None,
           "avg_launch_ms": round(avg_ms, 4), "stale": pm is None or tr is None}
    if pm is not None and secs > 0:
        simds = 1024.0                                   # 256 CUs x 4 SIMDs
        cycles = pm.get("GRBM_GUI_ACTIVE", 0.0) / 8.0    # the counter is summed over the 8 XCDs
        valu = pm.get("SQ_INSTS_VALU", 0.0)
        insts = sum(pm.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS",
                                               "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"))
        if cycles > 0 and valu > 0:
            clock_ghz = cycles / secs / 1e9              # the clock this run held, from the profiled cycle count
            out["achieved"] = round(valu / secs / 1e9, 2)
            out["peak"] = round(simds * clock_ghz / 2.0, 2)   # one wave64 VALU instruction per SIMD per two cycles
            out["frac"] = round(2.0 * valu / (cycles * simds), 4)
            out["issue"] = {"wave_instructions_per_launch": int(insts), "cycles_per_launch": int(cycles),
                            "valu_share": round(valu / insts, 3),
                            "salu_branch_share": round((pm.get("SQ_INSTS_SALU", 0.0) + pm.get("SQ_INSTS_BRANCH", 0.0)) / insts, 3),
                            "lds_share": round(pm.get("SQ_INSTS_LDS", 0.0) / insts, 3),
                            "cycles_per_instruction_per_simd": round(cycles * simds / insts, 3),
                            "resident_waves_per_simd": round(pm.get("SQ_WAVES", 0.0) / simds / pm.get("launches_added", 1), 2)}
            if pm.get("launches_added", 1) > 1:
                out["issue"]["launches"] = "k_shadow_lists + k_shadow (the light lists' own kernel, then the walk kernel for what it leaves): counters added"
    if tr is not None and secs > 0:
        hbm = tr.get("hbm_bytes_per_launch")
        if hbm:
            out["traffic"] = int(hbm)
            out["hbm_gbs"] = round(hbm / secs / 1e9, 1)
            out["hbm_frac"] = round(hbm / secs / 1e9 / HBM_PEAK_GBS, 4)
        req = tr.get("l2_requests_per_launch")
        if req:
            out["l2_gbs"] = round(req * 64.0 / secs / 1e9, 1)   # a vector-cache request to L2 carries 64 B
        if tr.get("l2_hit_rate") is not None:
            out["l2_hit_rate"] = round(tr["l2_hit_rate"], 4)
    alg_bytes = 64.0 * q + 8.0 * S + 40.0 * T
    out["flops"] = int(45 * full_tests)
    out["gflops"] = round(45.0 * full_tests / secs / 1e9, 2) if secs > 0 else None
    out["logical_gbs"] = round(alg_bytes / secs / 1e9, 1) if secs > 0 else None
    out["algorithmic_bytes_per_launch"] = int(alg_bytes)
    out["listed_tests_per_launch"] = int(T)
    out["executed_tests_frac"] = round(lane_slots / T, 4) if T else None
    out["source"] = ("launch time: HIP events of this run; SQ / L2 counters: rocprofv3 --pmc passes under profiles/ on the same "
                     "kernel sources (csrc_sha256), null when stale")
    out["note"] = ("frac is the share of SIMD cycles that issue a VALU instruction (the walk kernels run at the machine's issue rate, "
                   "about 2.1 cycles per instruction and SIMD: issue.cycles_per_instruction_per_simd says how far from it this one "
                   "is; the shadow queries answered from the light lists wait for memory and for their slowest wavefronts); HBM sees "
                   "hbm_frac of its peak; logical_gbs counts every listed triangle of the reference's walk as if read from memory "
                   "and is no fraction of a peak")
    return out


SCENES = {"temple": ("scenes/temple/temple", {}), "ulica": ("scenes/ulica/ulica", {}), "s3": ("scenes/s3/s3", {}),
          "s5": ("scenes/s5/s5", {}), "pokoj": ("scenes/pokoj/s2", {}), "s2": ("scenes/s2/s2", {})}


def run_leg(env, scene_name, width, height, steps, warmup, depth, latency_reps=3):
    """One workload on this job's ranks: a counting pass (rays per frame, S and T), `warmup` frames with HIP events around
    every kernel family, exactly `steps` timed frames between barriers, then `latency_reps` single frames start to finish.
    Returns the workload's result object on rank 0 (the members the top-level line and the config5 legs share), None
    on the other ranks."""
    torch, dist, rt, dev, rank, world, local_rank = (env[k] for k in ("torch", "dist", "rt", "dev", "rank", "world", "local_rank"))
    from raytracer_amd.dist import FramePipeline
    prefix, over = SCENES[scene_name]
    prefix = os.path.join(ROOT, prefix)
    scene = rt.Scene(prefix, width=width, height=height, **over)
    pipe = FramePipeline(lambda: rt.Renderer(local_rank), scene, rank, world, depth)
    renderer = pipe.renderers[0]

    # untimed counting pass: rays per frame, S and T per kernel family for the roofline (this rank's tiles)
    renderer.set_option("count", 1)
    with torch.cuda.stream(pipe.streams[0]):
        pipe.slots[0].frame()
    torch.cuda.synchronize(dev)
    cnt = renderer.counters().asdict()
    renderer.set_option("count", 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up frames carry HIP events around every kernel family (the per-kernel table of the result line); the timed
    # frames only around the two walk kernels, whose launch time the roofline object needs: the events of a whole frame
    # cost 0.07 ms of a 5.5 ms frame.
    nwarm = max(warmup, depth if depth > 1 else 0)
    renderer.set_option("timing", 1)
    renderer.timings()
    for _ in range(nwarm):
        pipe.frame()
    barrier()
    tim_warm = renderer.timings().asdict()
    warm_frames = (nwarm + depth - 1) // depth  # frames of the warm-up that ran in slot 0
    renderer.set_option("timing", 2)
    pipe.issued = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.frame()
    barrier()
    elapsed = time.perf_counter() - t0
    tim = renderer.timings().asdict()
    chunks = renderer.counters().asdict()
    timed_frames = (steps + depth - 1) // depth  # frames of the timed region that ran in slot 0, whose kernels are timed

    # one frame alone on the GPU(s), start to finish: what a caller that waits for every frame sees
    latency = None
    for _ in range(latency_reps):
        barrier()
        t1 = time.perf_counter()
        with torch.cuda.stream(pipe.streams[0]):
            pipe.slots[0].frame()
        barrier()
        dt = (time.perf_counter() - t1) * 1e3
        latency = dt if latency is None or dt < latency else latency

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rays_t = torch.tensor([cnt["nearest_queries"] + cnt["shadow_queries"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_frame = float(rays_t.item())
    pipe.close()
    del pipe
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    ms_step = elapsed / steps * 1e3
    fam = "shadow" if tim["shadow_ms"] >= tim["nearest_ms"] else "nearest"  # dominant kernel on this rank
    launches = max(tim[fam + "_launches"], 1)
    avg_ms = tim[fam + "_ms"] / launches
    if fam == "shadow":
        q, S, T = cnt["shadow_queries"], cnt["shadow_steps"], cnt["shadow_tests"]
    else:
        q, S, T = cnt["nearest_queries"], cnt["nearest_steps"], cnt["nearest_tests"]
    per_launch = timed_frames / launches  # frames per launch of this family (1 when the frame is one chunk and one level)
    full = cnt[fam + "_full_tests"]
    workload = "%s %dx%d" % (scene_name, width, height)
    roof = roofline(fam, avg_ms, q * per_launch, S * per_launch, T * per_launch, full * per_launch,
                    cnt[fam + "_lane_slots"] * per_launch, workload if world == 1 else None)
    roof["launches_per_frame"] = round(launches / max(timed_frames, 1), 2)

    def ratio(a, b):
        return round(a / b, 4) if b else None
    return {
        "value": round(rays_frame * steps / elapsed / 1e6, 3), "unit": "Mrays/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_step, 3), "frame_latency_ms": round(latency, 3) if latency is not None else None,
        "data": "bundled scene scenes/%s (the reference's own input), camera resolution set to %dx%d" % (scene_name, width, height),
        "config": {"workload": "scenes/%s %dx%d, depth 5, %d point lights, frame = preprocess + grid build + trace + "
                               "shade + tone map%s" % (scene_name, width, height, scene.nl,
                                                        " + RCCL min/max all-reduce + RGBA8 tile gather" if world > 1 else ""),
                   "rays_per_frame": int(rays_frame), "rays_per_pixel": round(rays_frame / (width * height), 2),
                   "tiling": "8x8 tiles round-robin over %d rank(s)" % world,
                   "frames_in_flight": depth, "hip_hardware_queues": int(os.environ.get("GPU_MAX_HW_QUEUES") or 4), "chunks_per_frame_rank0": chunks["chunks"], "chunks_rerendered_rank0": chunks["chunks_rerendered"]},
        "kernel_ms_per_frame_rank0": dict({k: round(v / max(warm_frames, 1), 3) for k, v in tim_warm.items() if k.endswith("_ms") and k not in ("preprocess_ms", "grid_ms")},
                                          **{k: round(tim[k] / timed_frames, 3) for k in ("nearest_ms", "shadow_ms")}),
        "setup_ms_on_build_stream_rank0": {k: round(tim_warm[k] / max(warm_frames, 1), 3) for k in ("preprocess_ms", "grid_ms")},
        "roofline": roof,
        "walk_stats_rank0": {
            f: {"simd_efficiency": ratio(cnt[f + "_lane_tests"], cnt[f + "_lane_slots"]),
                "wave_tests": cnt[f + "_lane_slots"] // 64, "wave_cell_visits": cnt[f + "_visits"],
                "batches_culled": ratio(cnt[f + "_batches_culled"], cnt[f + "_batches"]),
                "reached_second_slabs": ratio(cnt[f + "_stage2"], cnt[f + "_lane_slots"] / 64),
                "reached_distance_stage": ratio(cnt[f + "_stage3"], cnt[f + "_lane_slots"] / 64),
                "reached_exact_test": ratio(cnt[f + "_exact"], cnt[f + "_lane_slots"] / 64)}
            for f in ("nearest", "shadow")},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="temple")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames-in-flight", type=int, default=0, help="0 = for the headline workload 2 (3 from 8 GPUs on), and for the config5 legs 1 on one GPU, 2 on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-width", type=int, default=960)
    ap.add_argument("--cpu-height", type=int, default=540)
    ap.add_argument("--no-config5", action="store_true", help="skip the scenes/ulica legs (1920x1080 and 7680x4320) of the result line")
    args = ap.parse_args()

    # Before anything touches torch or the GPU: a plain `python bench.py --gpus N` starts its own ranks.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%s but --gpus %d" % (os.environ["WORLD_SIZE"], args.gpus))

    # Libraries below write banners to the C-level stdout (RCCL prints its version when a communicator is made); the
    # one JSON line is the only thing this program may print there, so stdout points to stderr until that line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # HIP maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, four by default; a new stream goes to the
    # queue with the fewest), and two streams on one queue run one after the other.  Two frames in flight keep four streams
    # busy (each context's and the one its next frame is set up on) beside the contexts' own streams, torch's default
    # stream and RCCL's: which of them share a queue depends on the order they were made in (temple 4K on one GPU, two
    # frames in flight: 1.96 ms per frame with four queues, 1.83 with eight, 1.80 with sixteen, 2.59 with two; the same
    # with the collectives of a group of one: 1.82 / 2.33 / 1.82).  Sixteen: a queue for every stream of the process.
    # Read when the HIP runtime starts, so set before torch is imported.
    if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < HW_QUEUES:
        os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES)

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.dist import default_depth, headline_depth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force = world == 1 and os.environ.get("RTGPU_BENCH_FORCE_COLLECTIVES") == "1"  # rehearsal of the N > 1 code on one GPU
    if world > 1 or force:
        if force:
            import socket
            with socket.socket() as sk:  # a port that is free now
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            import raytracer_amd.dist as rdist
            rdist.FORCE_COLLECTIVES = True
        dist.init_process_group("nccl", device_id=dev)

    env = {"torch": torch, "dist": dist, "rt": rt, "dev": dev, "rank": rank, "world": world, "local_rank": local_rank}
    # Frames in flight on this rank's GPU (FramePipeline): what dist.headline_depth says for the headline workload (two; three
    # from eight GPUs on: the latency-bound end of a frame's walk kernels runs under the next frames'), and what
    # dist.default_depth says for the config5 legs (one on a single GPU: ulica's frames fill it by themselves)
    depth = args.frames_in_flight if args.frames_in_flight > 0 else headline_depth(world)
    leg_depth = args.frames_in_flight if args.frames_in_flight > 0 else default_depth(world)
    main_leg = run_leg(env, args.scene, args.width, args.height, args.steps, args.warmup, depth)
    # ... and the same workload one frame at a time on a single GPU: a kernel's launch time is then its own (the counter
    # profiles under profiles/ are taken that way)
    alone_leg = None
    if world == 1 and depth > 1:
        try:
            alone_leg = run_leg(env, args.scene, args.width, args.height, args.steps, args.warmup, 1, 0)
        except Exception as e:
            alone_leg = {"error": str(e)}
    # Config 5 of the baseline (scenes/ulica at 7680x4320, maximum depth: the divergence stress) beside the headline
    # workload, with its own rooflines: 1920x1080 (three frames) and the full 7680x4320 (one frame), same ranks, same exchange.
    legs = {}
    if not args.no_config5 and args.scene == "temple":
        for name, w, h, k, wu, reps in (("ulica_1920x1080", 1920, 1080, 3, 1, 1), ("ulica_7680x4320", 7680, 4320, 1, 1, 0)):
            try:
                legs[name] = run_leg(env, "ulica", w, h, k, wu, leg_depth, reps)
            except Exception as e:  # an extra of the line: never a reason to lose the headline number
                legs[name] = {"error": str(e)}

    if rank == 0:
        out = {"metric": "Mrays/s (primary+secondary)", "value": main_leg["value"], "unit": "Mrays/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_leg["ms_per_step"],
               "frame_latency_ms": main_leg["frame_latency_ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": main_leg["data"], "config": main_leg["config"],
               "kernel_ms_per_frame_rank0": main_leg["kernel_ms_per_frame_rank0"],
               "setup_ms_on_build_stream_rank0": main_leg["setup_ms_on_build_stream_rank0"],
               "kernel_ms_note": "HIP events on the stream each kernel runs on: nearest_ms and shadow_ms over the timed frames, the others "
                                 "over the warm-up frames (the events of every kernel cost 0.07 ms per frame); with one frame in flight kernel_ms_per_frame_rank0 sums to the frame. "
                                 "setup_ms_on_build_stream_rank0 is the set-up of frame k+1 (preprocess, grid build: 0.27 ms alone) as it elapses on the build "
                                 "stream UNDER the walk kernels of frame k, where it waits for free compute units: not part of the frame's critical path"
                                 + ("; with %d frames in flight a kernel shares the GPU with the other frame's, so these are longer than "
                                    "the kernel alone" % depth if depth > 1 else ""),
               "roofline": main_leg["roofline"], "walk_stats_rank0": main_leg["walk_stats_rank0"]}
        if alone_leg is not None and "error" not in alone_leg:
            # The roofline object prices the dominant kernel by ITS OWN launch time: the K timed frames of the one-frame-at-a-time
            # leg of this run (the counter profiles and the kernel trace under profiles/ are taken the same way, --frames-in-flight 1).
            # Beside another frame's kernels the same launch lasts longer while the GPU does more per unit of time.
            roof = dict(alone_leg["roofline"])
            roof["avg_launch_ms_two_in_flight"] = main_leg["roofline"]["avg_launch_ms"]
            roof["timed_region"] = ("the %d timed frames of this run's one_frame_at_a_time leg (one frame in flight: the kernel has the GPU to "
                                    "itself); with %d frames in flight, as `value` is measured, the launch shares the GPU with the other frame's "
                                    "kernels and lasts avg_launch_ms_two_in_flight" % (alone_leg["steps"], depth))
            out["roofline"] = roof
        if alone_leg is not None:
            out["one_frame_at_a_time"] = alone_leg if "error" in alone_leg else {
                k: alone_leg[k] for k in ("value", "unit", "steps", "ms_per_step", "kernel_ms_per_frame_rank0", "setup_ms_on_build_stream_rank0", "roofline")}
        if legs:
            out["config5"] = legs
        if world == 1 and not args.no_cpu_baseline:
            prefix, over = SCENES[args.scene]
            try:
                out["cpu_baseline"] = cpu_baseline(os.path.join(ROOT, prefix), over, args.cpu_width, args.cpu_height)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": str(e)}
            if legs:  # config 5's baseline: the same scene at 160x90 (the reference needs 20 core-hours for a 7680x4320 frame)
                try:
                    out["config5"]["cpu_baseline"] = cpu_baseline(os.path.join(ROOT, SCENES["ulica"][0]), {}, 160, 90)
                except Exception as e:
                    out["config5"]["cpu_baseline"] = {"error": str(e)}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
