#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is 512 samples per pixel (`--spp-per-step`: two of the 256-spp renders of
the benchmark configurations, so that a step stays above 100 ms -- 4.3 G paths at
1080p) of the scene at 1920x1080, max depth 8, on every rank, issued as batches
of 32 samples per pixel (weak scaling: rank g of G renders
iterations g, g+G, ...).  With `--total-spp N` a step is instead one render of N
samples per pixel IN TOTAL, split over the ranks (strong scaling: the 4K
configurations with 1024 / 4096 spp).  The per-GPU radiance sums are combined
by one RCCL reduce (the library's own, `pt_reduce_sum`) at the end of the timed
region.  Rank 0 prints ONE JSON line; `value` is whole-job Mpaths/s.

The renderer keeps two batches in flight (two sets of path buffers, two HIP
streams), so the launches of consecutive steps overlap on the GPU.

Outside the timed region the same command also
  * checks the primary hit-ID buffer against the reference's (llvmpipe fixture),
  * re-runs the workload with every launch bracketed by HIP events on the stream it
    is launched on, once as timed (`roofline`: launches overlap) and once with one
    batch at a time (`roofline.alone`: every kernel alone on the GPU),
  * times the CPU oracle on a bounded sample of the same workload (rank 0).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (spec)
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak (spec, FMA = 2 flops)
# The intersection and shading arithmetic is compiled without contraction (one rounding per operation is the
# parity contract), so a multiply and an add are two instructions: the ceiling such code can reach is half the spec.
VALU_PEAK_UNCONTRACTED_TFLOPS = VALU_PEAK_TFLOPS / 2
# Algorithmic flops per work item: a node step tests two boxes (2 x (6 fma-equivalents + 5 min/max) = 44), the
# triangle test of pathtracer.frag.glsl:393-415 with n and dot(n, v0) hoisted is 85, finishing a segment (light
# test, hit point) 60, one BSDF sample with its frame, barycentrics and throughput update about 400 (the mean
# over the three lobes of frag:706-837), making a primary ray 60.
FLOPS_PER_NODE, FLOPS_PER_TRIANGLE, FLOPS_PER_SEGMENT, FLOPS_PER_SHADED_HIT, FLOPS_PER_PATH = 44.0, 85.0, 60.0, 400.0, 60.0
# VALU lane-operations per second the chip can issue: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12

# Algorithmic HBM bytes per work item with the queue records this build uses
# (three 16-byte lanes of path state, a 16-byte hit record, a 16-byte sample):
EXTEND_READ_PER_RAY = 32 + 4           # origin/direction lanes + light distance of every ray
EXTEND_WRITE_PER_HIT = 16              # hit-list record of a surface hit that will be shaded
SHADE_READ_PER_HIT = 16 + 48           # hit record + path state
SHADE_WRITE_PER_SURVIVOR = 48 + 4      # compacted state of a surviving path + its light distance
RAYGEN_BYTES_PER_RAY = 48 + 4          # the three lanes of a primary ray that is enqueued + its light distance (rays of pixels
                                       # outside the scene's window are never made: pt_stats.primary_rays_culled)
SAMPLE_BYTES = 16                      # one radiance sample per path whose primary ray was made (read by accumulate)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--scene", default="Box")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--depth", type=int, default=8)
    p.add_argument("--spp-per-step", type=int, default=512,
                   help="samples per pixel one step renders on every rank (weak scaling)")
    p.add_argument("--batch-spp", type=int, default=0,
                   help="samples per pixel in flight together (pt_config.iterations_per_batch); 0 = the renderer's own choice "
                        "for the scene and image size, reported in config.workload")
    p.add_argument("--batch-streams", type=int, default=2, choices=(1, 2),
                   help="2: two batches in flight on two streams (the product default); 1: one batch at a time")
    p.add_argument("--total-spp", type=int, default=0,
                   help="strong scaling: a step is one render of this many samples per pixel in total, rank g of G "
                        "rendering iterations g, g+G, ... of them")
    p.add_argument("--lattice", type=int, default=0,
                   help="instance the scene's shapes (all but its 10 room triangles) on an N^3 lattice")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-alone", action="store_true",
                   help="skip the pass that times every kernel alone on one stream (profiling runs: every "
                        "launch of the run is then of the overlapped kind)")
    p.add_argument("--no-extra", action="store_true", help="skip the secondary Veggie.obj measurement")
    p.add_argument("--wide-tree", action="store_true",
                   help="trees that are read from memory as four-wide nodes (pt_config.profile bit 5; measurement only)")
    p.add_argument("--native-reduce", action="store_true",
                   help="make the process group and the library's RCCL communicator even for one rank (what a "
                        "launch under torch.distributed.run does by itself)")
    return p.parse_args()


LATTICE = 0  # --lattice N: the scene's furniture instanced N^3 times in its room (N = 19: 9.88 M triangles)


def load_scene(pt, name):
    scene = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"))
    return scene.instance_lattice(10, LATTICE) if LATTICE else scene


def spp_of_step(args, rank, world):
    """Samples per pixel THIS rank renders in one step."""
    if args.total_spp:
        return max(0, (args.total_spp - rank + world - 1) // world)  # iterations rank, rank + world, ... < total
    return args.spp_per_step


class Reducer:
    """The one reduce at the end of a render.  The library's own RCCL path (pt_comm_*, pt_reduce_sum: ncclReduce
    on the renderer's stream); torch.distributed only carries the communicator id to the ranks."""

    def __init__(self, pt, dist, rank, world, device):
        from pathtracer_amd import distributed as ptd

        self.pt, self.kind = pt, "rccl (pt_reduce_sum)"
        self.comm = ptd.native_comm(rank, world, device)
        self.libraries = pt.comm_runtime_paths()  # which HIP runtime and which RCCL the reduce runs on

    def __call__(self, r):
        self.pt.reduce_sum([r], [self.comm], 0)

    def close(self):
        self.comm.close()


class ReduceFailed(RuntimeError):
    """The collective did not work on some rank (all ranks raise it together)."""


def timed_run(pt, dist, args, scene_name, rank, world, device, reducer):
    """Returns (seconds for K steps [max over ranks], renderer, samples per pixel the root holds per step)."""
    cam = pt.Camera(resolution=(args.width, args.height))
    r = pt.Renderer(load_scene(pt, scene_name), cam, max_depth=args.depth, device=device,
                    iterations_per_batch=args.batch_spp, batch_streams=args.batch_streams, profile=32 if args.wide_tree else 0)
    r.set_partition(rank, world)
    spp = spp_of_step(args, rank, world)
    for _ in range(args.warmup):
        r.draw(spp)
    if reducer:
        # the collective of the timed region once before it (connection set-up, kernel load) -- and the place to
        # find out that it does not work: a rank that failed alone inside the timed region would leave the others
        # waiting at the barrier, so the ranks agree here on whether it did
        why = ""
        try:
            reducer(r)
            r.synchronize()
        except Exception as e:  # noqa: BLE001
            why = f"{type(e).__name__}: {e}"
        if dist is not None:
            reasons = [None] * world
            dist.all_gather_object(reasons, why)
            why = next((w for w in reasons if w), "")
        if why:
            r.close()
            raise ReduceFailed(why)
    r.synchronize()
    r.reset()
    r.synchronize()

    def fence():
        r.synchronize()
        if dist is not None:
            dist.barrier()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.draw(spp)
        if reducer and args.total_spp:
            reducer(r)  # strong scaling: every step is a complete render, reduce included
    if reducer and not args.total_spp:
        reducer(r)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, r


def kernel_profile(pt, args, scene_name, device, rank, world, batch_spp, batch_streams=2, counters=True, separate_launches=False):
    """Same workload with every launch bracketed by HIP events on the stream it is
    launched on (pt_config.profile).  batch_streams = 2: as in the timed run, the
    kernels of two batches overlap and a launch's duration includes what it shares
    with the other batch; 1: every kernel alone on the GPU.  separate_launches: one
    launch per kernel and bounce also where the renderer would run all bounces of a
    batch in one launch (profile bit 4), for the per-kernel breakdown.  batch_spp: the batch size of the timed run
    (the profiling switches would change the renderer's own choice)."""
    cam = pt.Camera(resolution=(args.width, args.height))
    r = pt.Renderer(load_scene(pt, scene_name), cam, max_depth=args.depth, device=device,
                    iterations_per_batch=batch_spp, profile=1 | (16 if separate_launches else 0) | (32 if args.wide_tree else 0),
                    batch_streams=batch_streams)
    r.set_partition(rank, world)
    spp = max(1, spp_of_step(args, rank, world))
    r.draw(min(spp, batch_spp))  # warm
    r.synchronize()
    r.reset()
    steps = max(1, min(args.steps, 4))
    for _ in range(steps):
        r.draw(spp)
    st = r.stats()
    r.close()
    if not counters:
        return st
    # Work counters (BVH nodes fetched, triangle tests) come from a separate
    # instrumented pass so that they do not slow the timed kernels down.
    r = pt.Renderer(load_scene(pt, scene_name), cam, max_depth=args.depth, device=device,
                    iterations_per_batch=batch_spp, profile=2 | (32 if args.wide_tree else 0))
    r.set_partition(rank, world)
    r.draw(min(spp, batch_spp))
    c = r.stats()
    r.close()
    scale = st["segments"] / max(1, c["segments"])
    st["nodes_visited"] = int(c["nodes_visited"] * scale)
    st["tris_tested"] = int(c["tris_tested"] * scale)
    d = c["extend_diag"]  # [4] rounds of the traversal loop, [5] lanes holding a ray summed over them
    st["diag_lanes"] = d[5] / max(1, d[4])
    return st


def cpu_baseline(args, scene_name):
    """The CPU oracle (plain-C restatement of the reference shader, linear
    scan) on a bounded sample: a few iterations of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", scene_name + ".npz"))
    sc = ob.OracleScene({k: z[k] for k in z.files})
    import pathtracer_amd as pt

    cam = pt.Camera(resolution=(args.width, args.height))
    cam.update()
    oc = ob.make_camera(cam.eye, cam.forward, cam.up, cam.right, args.width, args.height)
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    n_it = 0
    while True:
        ob.render(sc, oc, n_it, n_it + 1, args.depth, threads)
        n_it += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n_it >= 64:
            break
    paths = args.width * args.height * n_it
    return {"value": paths / el / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": f"{scene_name}.obj {args.width}x{args.height} depth {args.depth}, iterations 0..{n_it - 1} "
                      f"({paths} paths, {el:.1f} s), oracle/pt_oracle.c with OpenMP"}


def main():
    global LATTICE
    args = parse_args()
    if args.lattice:
        # the CPU oracle scans every triangle per ray: no CPU leg on millions of triangles
        LATTICE, args.no_cpu_baseline, args.no_extra = args.lattice, True, True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # --gpus does not start processes: one rank per GPU comes from the launcher
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} ranks but WORLD_SIZE is {world}; launch it as\n"
                         f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port 29500 bench.py --gpus {args.gpus} --steps {args.steps} --warmup {args.warmup}")
    # one rank started by the launcher goes the whole multi-rank way too (process group, communicator, reduce)
    distributed = world > 1 or args.native_reduce or "TORCHELASTIC_RUN_ID" in os.environ
    dist = None
    # Rehearsal on a one-GPU box: every rank on GPU 0 (RCCL refuses two ranks on one device, so the sums are
    # combined on the host through gloo).  Exercises the multi-rank control flow only; its numbers mean nothing.
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # torch.distributed is the control plane only (communicator id, barrier, max over ranks of the time):
        # gloo on the host; the data path is the library's own RCCL reduce.
        import datetime

        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
        if rehearsal:
            local_rank = 0
    import pathtracer_amd as pt

    if pt.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    device = local_rank if distributed else 0

    reducer = None
    if distributed:
        import torch

        class HostReduce:
            """The sums through host memory and gloo: the rehearsal's reduce, and the fall-back when the library's
            RCCL communicator cannot be made on this machine (the line then says so in config.reduce)."""
            host_total = True

            def __init__(self, kind):
                self.kind, self.total = kind, None

            def __call__(self, r):
                t = torch.from_numpy(r.read_sum())
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
                self.total = t.numpy()

            def close(self):
                pass

        if rehearsal:
            reducer = HostReduce("gloo on the host (rehearsal)")
        else:
            why = ""
            try:
                reducer = Reducer(pt, dist, rank, world, device)
            except Exception as e:  # noqa: BLE001 -- whatever it is, the ranks must agree on what to do next
                why = f"{type(e).__name__}: {e}"
            ok = torch.tensor([1 if reducer is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if reducer is not None:
                    reducer.close()
                reasons = [None] * world
                dist.all_gather_object(reasons, why)
                why = next((w for w in reasons if w), "a rank failed")
                print(f"[bench] rank {rank}: no RCCL communicator ({why}); the sums go through the host", file=sys.stderr)
                reducer = HostReduce(f"gloo on the host (RCCL communicator failed: {why})")

    try:
        dt, r = timed_run(pt, dist, args, args.scene, rank, world, device, reducer)
    except ReduceFailed as e:
        # the communicator was made but the reduce itself failed: the sums go through the host, and the line says so
        print(f"[bench] rank {rank}: the RCCL reduce failed ({e}); the sums go through the host", file=sys.stderr)
        try:
            reducer.close()
        except Exception:  # noqa: BLE001
            pass
        reducer = HostReduce(f"gloo on the host (RCCL reduce failed: {e})")
        dt, r = timed_run(pt, dist, args, args.scene, rank, world, device, reducer)
    st = r.stats()
    n_pixels = args.width * args.height
    spp_all_ranks = args.total_spp if args.total_spp else args.spp_per_step * world  # per step, whole job
    total_paths = n_pixels * spp_all_ranks * args.steps
    value = total_paths / dt / 1e6

    gates = {}
    if rank == 0:
        fx = os.path.join(ROOT, "tests", "golden", "llvmpipe", f"hit_{args.scene}_{args.width}x{args.height}_it0.npz")
        if os.path.exists(fx) and not LATTICE:
            ids, _, _ = r.read_hit_ids(0)
            gates["primary_hit_ids_bit_exact_vs_reference"] = bool(np.array_equal(ids, np.load(fx)["ids"]))
        s = reducer.total if getattr(reducer, "host_total", False) else r.read_sum()
        gates["samples_per_pixel_root"] = float(s[..., 3].mean())
        gates["image_finite"] = bool(np.isfinite(s).all())
    r.close()

    batch_spp = st["iterations_per_batch"]  # what the timed renderer used (its own choice unless --batch-spp)
    prof = kernel_profile(pt, args, args.scene, device, rank, world, batch_spp, batch_streams=args.batch_streams)
    # the same kernels one at a time (one stream): what a launch takes when it has the GPU to itself
    prof_alone = None if args.no_alone else kernel_profile(pt, args, args.scene, device, rank, world, batch_spp, batch_streams=1,
                                                            counters=False)
    # where the timed run puts all bounces of a batch in one launch: the same work kernel by kernel, for the breakdown
    prof_sep = None
    if prof["launches_bounces"] and not args.no_alone:
        prof_sep = kernel_profile(pt, args, args.scene, device, rank, world, batch_spp, batch_streams=1, counters=False,
                                  separate_launches=True)
    out = None
    if rank == 0:
        seg = prof["segments"]

        def kernel_table(p):
            shaded = p["hits_shaded"]  # surface hits handed to the shade kernel
            culled = p.get("primary_rays_culled", 0)  # counted as segments, but never written nor read
            unlit = p.get("unlit_last_rays_skipped", 0)  # last-segment rays not aimed at the light: the same
            primaries = p["paths"] - culled
            extend_bytes = EXTEND_READ_PER_RAY * (p["segments"] - culled - unlit) + EXTEND_WRITE_PER_HIT * shaded
            shade_bytes = SHADE_READ_PER_HIT * shaded + SHADE_WRITE_PER_SURVIVOR * max(0, shaded - unlit)
            table = {
                "extend": {"ms_total": p["ms_extend"], "launches": p["launches_extend"],
                           "algorithmic_bytes": extend_bytes if p["launches_extend"] else 0},
                "shade": {"ms_total": p["ms_shade"], "launches": p["launches_shade"],
                          "algorithmic_bytes": shade_bytes if p["launches_shade"] else 0},
                # all bounces of a batch in one launch (k_bounces): the traversal and shading of every bounce
                "bounces": {"ms_total": p["ms_bounces"], "launches": p["launches_bounces"],
                            "algorithmic_bytes": (extend_bytes + shade_bytes + RAYGEN_BYTES_PER_RAY * primaries)
                            if p["launches_bounces"] else 0},  # the kernel also makes the primary rays
                "raygen": {"ms_total": p["ms_raygen"], "launches": p["launches_raygen"],
                           "algorithmic_bytes": RAYGEN_BYTES_PER_RAY * primaries if p["launches_raygen"] else 0},
                "accumulate": {"ms_total": p["ms_accumulate"], "launches": p["launches_accumulate"],
                               # (the samples of pixels whose primary rays were never made are not read)
                               "algorithmic_bytes": SAMPLE_BYTES * primaries + 32 * n_pixels * p["launches_accumulate"]},
            }
            for k in table.values():
                k["avg_launch_ms"] = k["ms_total"] / max(1, k["launches"])
                k["achieved_GBs"] = k["algorithmic_bytes"] / max(k["ms_total"], 1e-9) / 1e6
            return table

        kernels = kernel_table(prof)
        alone = kernel_table(prof_alone) if prof_alone else None
        dom = max(("extend", "shade", "bounces"), key=lambda n: kernels[n]["ms_total"])
        # Counters of the same command under rocprofv3 --pmc (separate passes, FETCH_SIZE doubled for gfx950),
        # kept with the profile summaries: HBM bytes per launch, VALU issue utilisation, active lanes.
        pmc, traffic = {}, None
        tag = f"{args.scene}_lattice{LATTICE}" if LATTICE else args.scene
        pmc_file = os.path.join(ROOT, "profiles", f"pmc_{tag}_{args.width}x{args.height}.json")
        if os.path.exists(pmc_file):
            allk = json.load(open(pmc_file))
            for name, m in allk.items():
                # the render-loop instantiation: <BLOCK, COUNT, EXACT, SCENE_LDS, RENDER, USE_LT> = no counters, no exact pass, render
                short = "extend" if re.fullmatch(r"k_extend_bvh<\d+, false, false, \d, true, (true|false)>", name) else \
                    ("shade" if name == "k_shade" else
                     ("bounces" if re.fullmatch(r"k_bounces<\d+, \d, (true|false)>", name) else None))
                if short:
                    pmc[short] = m
            if dom in pmc and "hbm_read_bytes_per_launch" in pmc[dom]:
                traffic = pmc[dom]["hbm_read_bytes_per_launch"] + pmc[dom].get("hbm_write_bytes_per_launch", 0.0)
        occupancy = {"shade_lane_fill": prof["hits_shaded"] / max(1, 64 * prof["shade_batches"]),
                     "extend_lanes_holding_a_ray": prof["diag_lanes"]}

        # ---- the dominant kernel against its two rooflines.
        # Work of ONE launch = the algorithmic bytes / flops per work item (the constants at the top of this file)
        # x the work items the kernels counted, / launches.  Duration = the kernel's average launch with the GPU
        # to itself (HIP events on its stream, one batch at a time); with two batches in flight a launch's
        # duration includes what it shares with the other batch's kernels and is not exclusive time.
        launches = max(1, kernels[dom]["launches"])
        flop_terms = {
            "nodes": prof["nodes_visited"] * FLOPS_PER_NODE, "triangle_tests": prof["tris_tested"] * FLOPS_PER_TRIANGLE,
            "segments": (seg - prof.get("primary_rays_culled", 0) - prof.get("unlit_last_rays_skipped", 0)) * FLOPS_PER_SEGMENT,
            "shaded_hits": prof["hits_shaded"] * FLOPS_PER_SHADED_HIT,
            "primary_rays": (prof["paths"] - prof.get("primary_rays_culled", 0)) * FLOPS_PER_PATH}
        in_kernel = {"extend": ("nodes", "triangle_tests", "segments"), "shade": ("shaded_hits",),
                     "bounces": ("nodes", "triangle_tests", "segments", "shaded_hits", "primary_rays")}[dom]
        flops_per_launch = sum(flop_terms[k] for k in in_kernel) / launches
        bytes_per_launch = kernels[dom]["algorithmic_bytes"] / launches
        exclusive_ms = alone[dom]["avg_launch_ms"] if alone and alone[dom]["launches"] else kernels[dom]["avg_launch_ms"]
        hbm_achieved = bytes_per_launch / max(exclusive_ms, 1e-9) / 1e6   # GB/s
        valu_achieved = flops_per_launch / max(exclusive_ms, 1e-9) / 1e9  # TFLOP/s
        hbm_view = {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
                    "bytes_per_launch": bytes_per_launch}
        valu_view = {"achieved": valu_achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_achieved / VALU_PEAK_TFLOPS,
                     "peak_uncontracted": VALU_PEAK_UNCONTRACTED_TFLOPS, "frac_uncontracted": valu_achieved / VALU_PEAK_UNCONTRACTED_TFLOPS,
                     "flops_per_launch": flops_per_launch,
                     "flops_per_path": {k: flop_terms[k] / max(1, prof["paths"]) for k in in_kernel}}
        nearer = "hbm" if hbm_view["frac"] >= valu_view["frac"] else "valu_fp32"
        top = hbm_view if nearer == "hbm" else valu_view
        # Why the kernel sits below both: it is bound by VALU *issue* with part of every instruction masked off
        # (while-while traversal of incoherent rays diverges inside the wave).  Share of issue cycles with a VALU
        # instruction in flight and active lanes per instruction come from the committed counter file, the
        # first rescaled to the launch duration measured here; their product is the share of the chip's VALU
        # lane-slots doing work -- a utilisation, not a roofline fraction (useless instructions would raise it).
        m = pmc.get(dom, {})
        gap = None
        if "valu_issue_utilisation" in m and "active_lanes_per_valu_instruction" in m:
            issue = m["valu_issue_utilisation"] * (m["avg_launch_us"] / 1e3) / max(exclusive_ms, 1e-9)
            lanes = m["active_lanes_per_valu_instruction"]
            gap = {"valu_issue_utilisation": issue, "lanes_active_of_64": lanes, "valu_lane_utilisation": issue * lanes / 64.0,
                   "valu_lane_ops_peak_T_per_s": VALU_PEAK_TLANEOPS,
                   "reading": f"{issue:.0%} of SIMD issue cycles carry a VALU instruction, {lanes:.1f} of 64 lanes active in it: "
                              f"{1.0 - lanes / 64.0:.0%} of every issued vector instruction is masked off by divergence",
                   "counters_from": os.path.relpath(pmc_file, ROOT)}
        roofline = {"kernel": dom, "bound": nearer, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
                    "frac": top["frac"], "traffic": traffic,
                    "definition": "algorithmic bytes (flops) of one launch / the launch's duration with the GPU to itself / peak; "
                                  "`bound` names the roofline the kernel is nearer to",
                    "avg_launch_ms": exclusive_ms, "avg_launch_ms_two_batches_in_flight": kernels[dom]["avg_launch_ms"],
                    "hbm": hbm_view, "valu_fp32": valu_view, "gap": gap,
                    "pmc": {"source": os.path.relpath(pmc_file, ROOT) if pmc else None, **pmc},
                    "lane_occupancy": occupancy,
                    # every kernel of the timed configuration (launches overlap) and with one batch at a time
                    "kernels": kernels, "alone": alone,
                    # the timed run puts all bounces of a batch in one launch; the same work as one launch per kernel
                    # and bounce, one batch at a time:
                    "as_separate_launches": kernel_table(prof_sep) if prof_sep else None}
        # The whole pipeline against HBM: algorithmic bytes of all kernels of a step over the step's wall time.
        bytes_per_path = sum(k["algorithmic_bytes"] for k in kernels.values()) / max(1, prof["paths"])
        pipeline_gbs = bytes_per_path * (total_paths / world) / dt / 1e9
        scene_label = f"{args.scene}.obj{f' instanced on a {LATTICE}^3 lattice' if LATTICE else ''}"
        out = {
            "metric": "Mpaths/sec at 1080p, 8 bounces, Box.obj & Veggie.obj; 1/2/4/8 MI355X",
            "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.total_spp else "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": f"{scene_label} of the reference's assets (committed dump of its Scene arrays), the reference's default "
                    "camera and light; samples are generated by the integrator's own RNG",
            "config": {"workload": f"{scene_label} {args.width}x{args.height}, max depth {args.depth}, "
                                   + (f"{args.total_spp} spp per step in total over {world} GPU(s)" if args.total_spp else
                                      f"{args.spp_per_step} spp per step per GPU")
                                   + f", batches of {batch_spp} spp, default camera and light",
                       "paths_per_step": n_pixels * spp_all_ranks, "parallelism": f"sample-partition x{world}",
                       "reduce": reducer.kind if reducer else None,
                       "reduce_libraries": getattr(reducer, "libraries", None)},
            "msegments_per_s": value * st["segments"] / max(1, st["paths"]),
            "segments_per_path": st["segments"] / max(1, st["paths"]),
            # primary rays of pixels outside the window the scene and the lights project to: counted as segments, never traced
            "primary_rays_culled_per_path": st.get("primary_rays_culled", 0) / max(1, st["paths"]),
            # rays of the last allowed segment that do not point at the light: counted as segments, never made
            "unlit_last_rays_skipped_per_path": st.get("unlit_last_rays_skipped", 0) / max(1, st["paths"]),
            "gates": gates,
            "roofline": roofline,
            "pipeline": {"algorithmic_bytes_per_path": bytes_per_path, "hbm_GBs_per_gpu": pipeline_gbs,
                         "hbm_frac": pipeline_gbs / HBM_PEAK_GBS},
            # outside the timed region: the tree and the uploads of pt_create
            "setup": {"triangles": st["n_triangles"], "bvh_nodes": st["n_nodes"], "bvh_depth": st["bvh_depth"],
                      "bvh_builder": "device" if st["bvh_builder"] else "host", "bvh_build_ms": st["bvh_build_ms"],
                      "upload_ms": st["upload_ms"]},
        }
        timing = os.path.join(ROOT, "tests", "golden", "llvmpipe", "llvmpipe_timing.json")
        if os.path.exists(timing):
            out["reference_llvmpipe_container"] = json.load(open(timing))
    if rank == 0 and not args.no_extra and args.scene != "Veggie" and not distributed:
        dt2, r2 = timed_run(pt, None, args, "Veggie", 0, 1, device, None)
        st2 = r2.stats()
        r2.close()
        out["veggie"] = {"value": n_pixels * spp_of_step(args, 0, 1) * args.steps / dt2 / 1e6, "unit": "Mpaths/s",
                         "n_gpus": 1, "ms_per_step": dt2 / args.steps * 1e3,
                         "segments_per_path": st2["segments"] / max(1, st2["paths"])}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, args.scene)
    if reducer:
        reducer.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
