#!/usr/bin/env python3
"""Times the device BVH build on the instanced 10-million-triangle scene (or --lattice N).

    python tools/bvh_device_time.py [--lattice 19] [--repeat 3] [--host]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, default=19)
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--host", action="store_true")
args = ap.parse_args()
scene = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "Veggie.npz"))
if args.lattice:
    scene = scene.instance_lattice(10, args.lattice)
for builder in (["host"] if args.host else []) + ["device"] * args.repeat:
    t0 = time.perf_counter()
    r = pt.Renderer(scene, pt.Camera(resolution=(64, 64)), max_depth=8, bvh_builder=builder)
    dt = time.perf_counter() - t0
    st = r.stats()
    r.close()
    print(f"{builder}: {st['n_triangles']} triangles, {st['n_nodes']} nodes, depth {st['bvh_depth']}: "
          f"build {st['bvh_build_ms']:.1f} ms, upload {st['upload_ms']:.1f} ms, pt_create {1e3 * dt:.0f} ms", flush=True)
