#!/usr/bin/env python3
"""One-GPU throughput of every benchmark configuration (bounded sample counts).

    python tools/bench_configs.py [--out gpurun_out/configs.json] [--only NAME]

Each line: scene, resolution, depth, samples rendered, Mpaths/s, segments per
path and the per-launch averages of the extend and shade kernels (HIP events,
second pass).  The 4K / 10M-triangle cases are the multi-GPU configurations run
on one rank's share of the samples.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402

CONFIGS = [
    # name, scene, lattice n, width, height, depth, samples per pixel of a run = these two multiplied (the batch size is
    # the renderer's own choice for the scene and image size)
    ("box_512_d5", "Box", 0, 512, 512, 5, 64, 4),
    ("box_1080p_d8", "Box", 0, 1920, 1080, 8, 32, 6),
    ("veggie_1080p_d8", "Veggie", 0, 1920, 1080, 8, 32, 6),
    ("transmission_1080p_d8", "Box_Transmission", 0, 1920, 1080, 8, 32, 6),
    ("transmission_sphere_1080p_d8", "Box_TransmissionSphere", 0, 1920, 1080, 8, 32, 6),
    ("anisotropic_1080p_d8", "Box_Anisotropic", 0, 1920, 1080, 8, 32, 6),
    ("anisotropic_test_1080p_d8", "AnisotropicTest", 0, 1920, 1080, 8, 32, 6),
    ("veggie_4k_d8", "Veggie", 0, 3840, 2160, 8, 8, 6),
    ("veggie_lattice10M_4k_d8", "Veggie", 19, 3840, 2160, 8, 8, 4),
]


def run(name, scene_name, lattice, w, h, depth, spp, batches):
    scene = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", scene_name + ".npz"))
    t0 = time.perf_counter()
    if lattice:
        scene = scene.instance_lattice(10, lattice)
    cam = pt.Camera(resolution=(w, h))
    r = pt.Renderer(scene, cam, max_depth=depth)
    total = 2 * spp * batches
    spp = r.stats()["iterations_per_batch"]
    batches = max(2, total // spp)
    out = {"name": name, "scene": scene_name, "triangles": scene.n_triangles, "width": w, "height": h,
           "depth": depth, "spp": spp * batches, "batch_spp": spp}
    out["setup_s"] = time.perf_counter() - t0
    r.draw(2 * spp)  # one batch on each of the renderer's two streams
    r.synchronize()
    dt = float("inf")
    for _ in range(3):  # best of three: a configuration here is tens of milliseconds long
        r.reset()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(batches):
            r.draw(spp)
        r.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    st = r.stats()
    r.close()
    out["mpaths_per_s"] = w * h * spp * batches / dt / 1e6
    out["segments_per_path"] = st["segments"] / st["paths"]
    out["mrays_per_s"] = st["segments"] / dt / 1e6
    out["stack_entries"] = st["stack_entries"]
    r = pt.Renderer(scene, cam, max_depth=depth, iterations_per_batch=spp, profile=1 | 16)  # per-kernel times: one launch per kernel
    r.draw(spp)
    r.synchronize()
    r.reset()
    r.draw(spp * min(batches, 2))
    st = r.stats()
    r.close()
    out["extend_us_per_launch"] = 1e3 * st["ms_extend"] / max(1, st["launches_extend"])
    out["shade_us_per_launch"] = 1e3 * st["ms_shade"] / max(1, st["launches_shade"])
    r = pt.Renderer(scene, cam, max_depth=depth, iterations_per_batch=1, profile=2)
    r.draw(1)
    st = r.stats()
    r.close()
    out["nodes_per_ray"] = st["nodes_visited"] / st["segments"]
    out["tris_per_ray"] = st["tris_tested"] / st["segments"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    rows = []
    for cfg in CONFIGS:
        if args.only and args.only not in cfg[0]:
            continue
        row = run(*cfg)
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
