#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[2] ("Msamples/s ... 1M-tri"): procedural 1M-triangle soup in the 5-wall box
with one quad light AND a procedural 2048x1024 sky as an importance-sampled environment-map light (an extension: the
reference has only a constant background — `--no-envmap` gives the reference's own feature set), 1920x1080, 256
samples per pixel per GPU, max_depth 50, no Russian roulette (the reference's estimator).  A "step" is one full
render: camera rays -> ... -> framebuffer in HBM (-> one RCCL gather of the strips to rank 0 when N > 1).  Scene
build/upload happen before the timed region; the scene, the path state and the framebuffer are resident in HBM.

Scaling: the default is weak — every GPU renders 1920x1080x256 samples' worth of work (with N GPUs the image keeps its
size, the rows are sharded in 4-row strips over the ranks and the sample count per pixel is 256*N); the `metric`
string says so.  `--config 3` is BASELINE configs[3] — the north-star's ">= 6x at 8 GPUs" statement: the same scene at
4096x4096 with 1024 spp TOTAL, rows sharded over the N ranks (strong scaling):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8 --config 3
For N > 1 the line carries `multi_gpu`: backend, world size as the process group reports it, every rank's render and
gather milliseconds.

Precision: `value` is measured on the MIXED-precision device path (TAKE_PRECISION_MIXED): every path's first three
rounds — camera ray and the next two bounces, where a flipped hit / miss decision costs the most radiance — run in the
reference's arithmetic (Real = double, src/take.h:27) on the f64 scene, the surviving paths continue on f32 records and
the f32 scene.  The line carries its own proof: at N = 1 the same workload is also timed on the pure f64 path
(`reference_precision`: images at rounding level of the pinned oracle, tests/test_gpu_parity.py) and on the pure f32
path (`production_f32`), and `parity` reports the per-pixel RMSE of the mixed image and of the f32 image against the
f64 image of this very workload at matched seeds (north-star tolerance: 1e-3; f32 alone sits at ~1.8e-3 on the 1M
soup, the mixed path at ~0.6e-3).  `--precision f64` / `f32` make one of the pure paths the headline.

One JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel
(closest-hit traversal, measured with HIP events inside the timed region) and `cpu_baseline` (rank 0, N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
STATE_BYTES_PER_RAY = 4 + 6 * 4 + 4 * 4  # queue word + origin/direction read + hit record written (f32)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=None, help="default 1920 (config 2) / 4096 (config 3)")
    ap.add_argument("--height", type=int, default=None, help="default 1080 / 4096")
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel: per GPU, default 256 (config 2); total, default 1024 (config 3)")
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--spb", type=int, default=0, help="samples per pixel per batch (0 = auto)")
    ap.add_argument("--materials", default="diffuse", choices=["diffuse", "mixed"],
                    help="mixed = round-robin over 7 BSDFs (configs[4]'s divergence stress; not the default workload)")
    ap.add_argument("--no-envmap", dest="envmap", action="store_false",
                    help="drop the procedural 2048x1024 sky (configs[2]'s importance-sampled env-map IBL, on by default; "
                         "an extension — the reference has only a constant background)")
    ap.add_argument("--instanced", default="", metavar="NxT",
                    help="configs[4]: N placements of a T-triangle mesh, mixed BSDFs (e.g. 1000x10000), traversed on two "
                         "levels (TakeInstance: one prototype + N transforms); replaces the soup (not the default workload)")
    ap.add_argument("--flatten", action="store_true",
                    help="with --instanced: scene_create expands the placements to world-space triangles (what the reference's "
                         "scene model can express; the instanced render is specified to equal it) — 40x the memory, a third faster")
    ap.add_argument("--builder", default="host", choices=["host", "device", "auto"],
                    help="host = binned SAH (default: the best trees); device = records + LBVH built on the GPU (fast "
                         "build, 2-6 %% slower traversal); auto = the library's default (host below 4M shapes)")
    ap.add_argument("--max-leaf", type=int, default=0, help="primitives per BVH leaf (0 = builder default)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE.json configs index: 2 = 1920x1080x256 spp per GPU (weak scaling, default); 3 = 4096x4096, "
                         "1024 spp total, rows sharded over the ranks (strong scaling)")
    ap.add_argument("--precision", default="mixed", choices=["f32", "f64", "mixed"],
                    help="arithmetic of the timed steps (`value`): mixed = first --exact-bounces rounds in f64, the rest in f32 "
                         "(default: meets the RMSE tolerance, see `parity`); f64 = the reference's Real throughout; f32 = the "
                         "production arithmetic throughout")
    ap.add_argument("--exact-bounces", type=int, default=0, help="mixed precision: rounds computed in f64 (0 = the library's default, 3)")
    ap.add_argument("--alt-steps", "--f64-steps", dest="alt_steps", type=int, default=2,
                    help="timed steps of the same workload on each of the OTHER precisions' device paths, N = 1 only; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="640x360x1", help="WxHxSPP of the CPU-baseline sample")
    args = ap.parse_args()
    dw, dh, ds = (4096, 4096, 1024) if args.config == 3 else (1920, 1080, 256)
    args.width = dw if args.width is None else args.width
    args.height = dh if args.height is None else args.height
    args.spp = ds if args.spp is None else args.spp
    return args


def measured_counters(args, precision):
    """What the committed rocprofv3 passes say about the dominant kernel of THIS command at `precision`.  PMC counters
    cannot be read from inside a run: the figures come from profiles/rNN_traffic.json (tools/profile_default.sh ->
    tools/profile_summary.py: FETCH_SIZE and WRITE_SIZE in separate passes, corrected with the factor
    tools/ubench_gather measures for this access pattern; SQ counters in their own pass).  Returns a dict with
    `traffic` (fabric bytes per launch or None), `valu_busy`, `source`; empty values when the run is not the default
    configuration the profile was taken on."""
    default = (args.tris, args.width, args.height, args.spp, args.max_depth, args.spb, args.materials, args.builder,
               args.max_leaf, args.envmap, args.instanced, args.config) == (
                   1_000_000, 1920, 1080, 256, 50, 0, "diffuse", "host", 0, True, "", 2)
    none = {"traffic": None, "valu_busy": None, "l2_hit_rate": None}
    if not (default and args.gpus == 1):
        return dict(none, source="not measured for this configuration (PMC passes exist for the default N = 1 command only)")
    for name in ("r03_traffic.json", "r02_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            t = json.load(f)
        t = t.get(precision, t if (precision == "f32" and "hbm_bytes_per_launch" in t) else None)
        if not t:
            continue
        return {"traffic": t.get("hbm_bytes_per_launch_corrected", t.get("hbm_bytes_per_launch")),
                "valu_busy": t.get("valu_busy"), "l2_hit_rate": t.get("l2_hit_rate"),
                "source": f"profiles/{name}: committed rocprofv3 --pmc passes of this command ({precision} leg), not this run"}
    return dict(none, source="no committed PMC profile for this precision")


def usable_cpus():
    """host cores this process may actually use: affinity mask, cgroup v2 quota, and the GPU box's stated share
    (16 CPUs per GPU) — os.cpu_count() reports the whole 256-thread host."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, sd_full):
    """Time the CPU path on a bounded sample of the same workload (same scene, fewer pixels / samples).
    Prefers the compiled reference itself (oracle/_ref/ref_harness, built in the authoring container);
    otherwise the oracle restatement.  Checker code, used here only as the thing the GPU is compared with."""
    from take_amd import scenes

    w, h, spp = (int(x) for x in args.cpu_sample.split("x"))
    cores = usable_cpus()
    sd = scenes.soup_scene(args.tris, w, h, spp=spp, max_depth=args.max_depth, materials=args.materials,
                           envmap=(2048, 1024) if args.envmap else None)
    samples = w * h * spp
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    sample = f"{args.tris}-tri soup, {w}x{h}, {spp} spp, max_depth {args.max_depth}"
    if os.path.exists(harness):
        # the reference has no environment-map light: it renders the same scene with its constant background
        sd_ref = scenes.soup_scene(args.tris, w, h, spp=spp, max_depth=args.max_depth, materials=args.materials)
        with tempfile.TemporaryDirectory() as td:
            xml = scenes.write_reference_inputs(sd_ref, td)
            r = subprocess.run([harness, "time", xml, str(args.max_depth), str(cores)], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=1500)
        secs = None
        for line in r.stdout.splitlines():
            if "Finish building rendering. Took" in line:  # the reference's own timer, src/render.cpp:83
                secs = float(line.split("Took")[1].split()[0])
        if r.returncode == 0 and secs:
            return {"value": samples / secs / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "reference",
                    "sample": sample + " (reference render() timer, seed-patched build"
                              + ("; without the env-map light, which the reference does not have)" if args.envmap else ")")}
    import oracle

    osc = oracle.OracleScene(sd, precision=1)
    osc.render(spp, args.max_depth, rng_mode=oracle.RNG_MT_PER_TILE, threads=cores)
    secs = osc.seconds
    osc.close()
    return {"value": samples / secs / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": sample + " (oracle <double, mt19937> tile loop)"}


KERNEL_MS = ("ms_trace_closest", "ms_trace_shadow", "ms_shade", "ms_other", "ms_total")


def roofline_of(args, precision, bytes_per_ray, acc, node_bytes):
    """`roofline` of the closest-hit kernel of one leg: achieved = algorithmic bytes per launch / average launch time
    (HIP events on the render stream, inside the timed region); the committed PMC passes add what reaches the fabric
    and how busy the vector ALUs are."""
    # (mixed precision: the dominant kernel is the f32 instance, which runs all rounds but the first three — its own
    # launches, its own time, its own rays)
    tail = precision == "mixed" and acc.get("launches_trace_closest_f32", 0) > 0
    n_launch = max(acc["launches_trace_closest_f32" if tail else "launches_trace_closest"], 1)
    avg_ms = acc["ms_trace_closest_f32" if tail else "ms_trace_closest"] / n_launch
    rays_in = acc["rays_closest_f32" if tail else "rays_closest"]
    bytes_per_launch = bytes_per_ray * rays_in / n_launch
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    pmc = measured_counters(args, precision)
    fabric = pmc["traffic"] / (avg_ms * 1e-3) / 1e9 if (pmc["traffic"] and avg_ms > 0) else None
    rtype = "double" if precision == "f64" else "float"  # (mixed: the f32 instance runs all rounds but the first three)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc["traffic"], "traffic_source": pmc["source"],
            "fabric_GBps": fabric, "fabric_frac": fabric / HBM_PEAK_GBS if fabric else None,
            "valu_busy": pmc["valu_busy"], "l2_hit_rate": pmc["l2_hit_rate"],
            "limiter": "not HBM: VALU issue (valu_busy) together with the vector L1's gather rate (one look-up per lane "
                       "and 16-byte load: tools/ubench_gather) and the fabric's random-request rate; the nodes and "
                       "records (~100 MB) live in L2 / Infinity Cache",
            "achieved_is": "ALGORITHMIC bytes (node_visits x node bytes + prim_tests x record bytes + ray state) per launch "
                           "/ launch time: mostly served by L2 / Infinity Cache, which is why it can approach the HBM "
                           "peak; `traffic` = FETCH_SIZE + WRITE_SIZE per launch = what reaches the fabric — FETCH_SIZE "
                           "counts Infinity-Cache hits too, so `fabric_frac` is an upper bound of the DRAM share",
            "kernel": f"tk::k_trace_group<{rtype},1,false,false,PathIo<{rtype}>,true> (closest hit, one ray per lane, "
                      "64-byte compressed nodes)",
            "launches": n_launch, "avg_launch_ms": avg_ms, "bytes_per_ray": bytes_per_ray, "node_bytes": node_bytes,
            "rays_per_launch": rays_in / n_launch}


def run_leg(args, sd, precision, steps, warmup, rank, world, backend, spp_total, is_main):
    """Build the scene at `precision`, warm up, time `steps` steps (render + gather when N > 1).  Returns a dict:
    value, elapsed, per-kernel ms, the last image (rank 0), roofline inputs.  The timed region is bracketed by a
    barrier + synchronize on both sides and the maximum over the ranks is taken."""
    import torch
    import torch.distributed as dist

    from take_amd import capi
    from take_amd import cdefs as D
    from take_amd.dist import gather_strips, multi_gpu_report, strip_rows

    f64 = precision != "f32"  # (records and images of the mixed path are doubles)
    t0 = time.time()
    scene = capi.Scene(sd, precision={"f32": D.TAKE_PRECISION_F32, "f64": D.TAKE_PRECISION_F64, "mixed": D.TAKE_PRECISION_MIXED}[precision],
                       max_leaf_size=args.max_leaf,
                       builder={"device": D.TAKE_BUILDER_DEVICE_LBVH, "host": D.TAKE_BUILDER_HOST_SAH, "auto": D.TAKE_BUILDER_AUTO}[args.builder],
                       flatten_instances=bool(args.instanced) and args.flatten)
    t_setup = time.time() - t0
    scene.exact_bounces = args.exact_bounces
    stats = scene.stats()
    rows = strip_rows(args.height, rank, world)
    out = torch.empty((len(rows), args.width, 3), dtype=torch.float64 if f64 else torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def render(spp, depth, seed, spb):
        scene.render_device(out.data_ptr(), spp, depth, seed=seed, strip_first=rank, strip_stride=world,
                            samples_per_batch=spb, stream=stream)

    # algorithmic bytes per closest-hit ray: one counting pass (instrumented kernel, never timed), 1 spp
    scene.set_instrumentation(timing=False, counting=True)
    render(1, args.max_depth, 0, 0)
    cc = scene.counters()
    rays_counted = cc["rays_closest"] + cc["rays_shadow"]
    bytes_per_ray = ((cc["node_visits"] * cc["node_bytes"] + cc["prim_tests"] * cc["prim_bytes"]) / max(rays_counted, 1)
                     + (2 * STATE_BYTES_PER_RAY - 4 if precision == "f64" else STATE_BYTES_PER_RAY))
    scene.set_instrumentation(timing=True, counting=False)
    if warmup == 0 and not is_main:
        # (secondary leg: kernels loaded AND the batch workspace of the timed steps allocated — a first hipMalloc of
        # > 100 GB of path state costs seconds and is not the workload; depth 0 keeps it to two rounds)
        render(spp_total, 0, 7, args.spb)
        render(1, args.max_depth, 7, 1)
    for _ in range(warmup):
        render(spp_total, args.max_depth, 0, args.spb)
        if world > 1:
            gather_strips(out, args.height, rank, world)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    acc = {"ms_trace_closest": 0.0, "launches_trace_closest": 0, "rays_closest": 0, "rays_shadow": 0,
           "ms_trace_shadow": 0.0, "ms_shade": 0.0, "ms_other": 0.0, "ms_total": 0.0, "bounces": 0,
           "rays_closest_f32": 0, "ms_trace_closest_f32": 0.0, "launches_trace_closest_f32": 0}
    img = None
    render_s = gather_s = 0.0
    for _ in range(steps):
        ta = time.perf_counter()
        render(spp_total, args.max_depth, 0, args.spb)  # (returns when the stream has drained)
        tb = time.perf_counter()
        img = out
        if world > 1:
            img = gather_strips(out, args.height, rank, world)
            torch.cuda.synchronize()
        tc = time.perf_counter()
        render_s += tb - ta
        gather_s += tc - tb
        c = scene.counters()
        for k in acc:
            acc[k] += c[k]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    report = multi_gpu_report(render_s / max(steps, 1) * 1e3, gather_s / max(steps, 1) * 1e3) if world > 1 else None
    if rank == 0:
        assert img is not None and torch.isfinite(img).all()
    keep = rank == 0 and world == 1 and img is not None and (not is_main or args.alt_steps > 0)  # (for `parity`)
    image64 = img.to(torch.float64).clone() if keep else None
    scene.close()
    del out
    torch.cuda.empty_cache()
    samples = args.width * args.height * spp_total * steps
    return {"precision": precision, "value": samples / elapsed / 1e6, "elapsed": elapsed, "steps": steps, "acc": acc,
            "bytes_per_ray": bytes_per_ray, "node_bytes": cc["node_bytes"], "stats": stats, "setup_s": t_setup,
            "image64": image64, "multi_gpu": report, "samples": samples}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from take_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    backend = os.environ.get("TAKE_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on a 1-GPU box
    dev = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    strong = args.config == 3
    spp_total = args.spp if strong else args.spp * world  # weak scaling: per-GPU samples fixed; configs[3]: total fixed
    if args.instanced:
        n_inst, n_tri = (int(x) for x in args.instanced.split("x"))
        # (--flatten: the library expands the placements itself, TakeBuildOpts.instances = TAKE_INSTANCES_FLATTEN)
        sd = scenes.instanced_scene(n_inst, n_tri, args.width, args.height, spp=spp_total, max_depth=args.max_depth)
        if args.envmap:
            sd.add_envmap(scenes.sky_envmap(2048, 1024))
    else:
        sd = scenes.soup_scene(args.tris, args.width, args.height, spp=spp_total, max_depth=args.max_depth,
                               materials=args.materials, envmap=(2048, 1024) if args.envmap else None)

    main_leg = run_leg(args, sd, args.precision, args.steps, args.warmup, rank, world, backend, spp_total, True)
    if rank == 0:
        acc, stats = main_leg["acc"], main_leg["stats"]
        mode = ("strong scaling: BASELINE configs[3], 4096x4096 x 1024 spp total, rows sharded over the ranks" if strong
                else f"weak scaling: {args.spp} spp per GPU")
        if args.instanced:
            geometry = f"{args.instanced} instanced triangles (configs[4]'s shape)"
        else:
            geometry = "1M-tri soup" if args.tris == 1_000_000 else f"{args.tris}-tri soup"
        line = {
            "metric": f"Msamples/s (camera paths/s: rays traced x spp / s), {geometry}; {mode}",
            "value": main_leg["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_leg["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": {"mixed": "f64+f32"}.get(args.precision, args.precision), "data": "synthetic",
            "config": {"workload": (f"{args.instanced} placements x triangles ({'flattened to world space' if args.flatten else 'two-level instancing'}, 7 BSDFs round-robin)" if args.instanced else
                                    f"procedural {args.tris}-triangle soup ({args.materials} materials)")
                                   + " in 5-wall box + 1 quad area light, "
                                   f"{args.width}x{args.height}, "
                                   + (f"{spp_total} spp total (BASELINE configs[3], rows sharded over the ranks)" if strong
                                      else f"{args.spp} spp per GPU ({spp_total} total)") + ", max_depth "
                                   f"{args.max_depth}, no Russian roulette, "
                                   + ("procedural sky env-map 2048x1024, importance-sampled (extension)" if args.envmap
                                      else "constant background (no env-map IBL upstream)"),
                       "arithmetic": {"f64": "f64 = the reference's Real (src/take.h:27); images at rounding level of the pinned oracle",
                                      "f32": "f32 production arithmetic (see parity)",
                                      "mixed": f"mixed: rounds 0..{(args.exact_bounces or 3) - 1} of every path (camera ray + next bounces) in f64 on "
                                               "the f64 scene, the rest on f32 records and the f32 scene; images f64 (see parity: "
                                               "within the tolerance of the all-f64 image of this workload)"}[args.precision],
                       "parallelism": f"tile-row strips over {world} GPU(s), scene replicated, one gather",
                       "bvh": {"builder": args.builder, "nodes": stats["n_nodes"], "prims": stats["n_prims"], "depth": stats["depth"],
                               "scene_bytes": stats["device_bytes"]},
                       "setup_s": main_leg["setup_s"], "rays_per_sample": (acc["rays_closest"] + acc["rays_shadow"]) * world
                       / max(main_leg["samples"], 1),
                       "kernel_ms": {k: acc[k] for k in KERNEL_MS}},
            "roofline": roofline_of(args, args.precision, main_leg["bytes_per_ray"], acc, main_leg["node_bytes"]),
        }
        if world > 1:
            line["multi_gpu"] = main_leg["multi_gpu"]
        if world == 1 and args.alt_steps > 0:
            # the same workload on the other arithmetics (own scene, own warm-up), and every leg's image against the f64 one
            images = {args.precision: main_leg["image64"]}
            for alt_prec in [p for p in ("f64", "f32", "mixed") if p != args.precision and not (p == "mixed" and args.precision != "mixed")]:
                alt = run_leg(args, sd, alt_prec, args.alt_steps, 0, rank, world, backend, spp_total, False)
                images[alt_prec] = alt["image64"]
                leg = {"dtype": alt_prec, "value": alt["value"], "unit": "Msamples/s", "steps": alt["steps"],
                       "ms_per_step": alt["elapsed"] / alt["steps"] * 1e3, "setup_s": alt["setup_s"],
                       "kernel_ms": {k: alt["acc"][k] for k in KERNEL_MS},
                       "roofline": roofline_of(args, alt_prec, alt["bytes_per_ray"], alt["acc"], alt["node_bytes"])}
                if alt_prec == "f32":
                    leg["note"] = ("faster, but its image is parity.f32_vs_f64_rmse away from the reference-precision image of "
                                   "this workload (north-star tolerance: 1e-3) — reported, not the headline")
                line["production_f32" if alt_prec == "f32" else "reference_precision"] = leg
            ref = images.get("f64")
            if ref is not None:
                par = {"spp": args.spp, "seed": 0, "tolerance": 1e-3,
                       "note": "per-pixel RMSE between the images of the timed steps of each leg and the f64 leg's (same workload, "
                               "matched counter seeds); the f64 device path agrees with the pinned oracle at rounding level "
                               "(tests/test_gpu_parity.py), so these are distances from the reference's arithmetic"}
                for name, im in images.items():
                    if name == "f64" or im is None:
                        continue
                    d = (im - ref).abs().amax(dim=2)
                    par[f"{name}_vs_f64_rmse"] = float(((im - ref) ** 2).mean().sqrt())
                    par[f"{name}_pixels_within_1e-3"] = float((d < 1e-3).double().mean())
                    par[f"{name}_mean_rel_diff"] = float(((im.mean() - ref.mean()) / ref.mean()).abs())
                if args.precision != "f64":
                    par["headline_within_tolerance"] = bool(par.get(f"{args.precision}_vs_f64_rmse", 1.0) < 1e-3)
                line["parity"] = par
        if world == 1 and not args.no_cpu_baseline and not args.instanced:
            try:
                line["cpu_baseline"] = cpu_baseline(args, sd)
            except Exception as e:  # the bench line must still come out
                line["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
