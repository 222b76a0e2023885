#!/usr/bin/env python3
"""Renders the BVH-stress configuration (Veggie.obj instanced on a 19^3 lattice,
9.9M triangles, 3840x2160, depth 8) for a few samples per pixel; meant to be
run under rocprofv3.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "Veggie.npz")).instance_lattice(10, 19)
r = pt.Renderer(scene, pt.Camera(resolution=(3840, 2160)), max_depth=8, iterations_per_batch=4)
r.draw(4); r.synchronize(); r.reset(); r.synchronize()
t0 = time.perf_counter()
r.draw(spp); r.synchronize()
dt = time.perf_counter() - t0
st = r.stats(); r.close()
print(json.dumps({"triangles": scene.n_triangles, "spp": spp, "mpaths_per_s": 3840 * 2160 * spp / dt / 1e6,
                  "mrays_per_s": st["segments"] / dt / 1e6, "segments_per_path": st["segments"] / st["paths"],
                  "bvh_nodes": st["n_nodes"], "stack_entries": st["stack_entries"]}))
