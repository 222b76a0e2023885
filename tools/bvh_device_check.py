#!/usr/bin/env python3
"""Builds the BVH of a scene with the host and the device builder and compares the trees.

    python tools/bvh_device_check.py [--lattice N ...] [--scenes Box Veggie ...]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402


def build(scene, builder):
    cam = pt.Camera(resolution=(64, 64))
    t0 = time.perf_counter()
    r = pt.Renderer(scene, cam, max_depth=8, bvh_builder=builder)
    dt = time.perf_counter() - t0
    st = r.stats()
    nodes, tris = r.read_bvh()
    ids = r.read_hit_ids(0)[0]
    r.close()
    return st, nodes, tris, ids, dt


def compare(name, scene):
    sh, nh, th, ih, dth = build(scene, "host")
    sd, nd, td, idd, dtd = build(scene, "device")
    ok = True
    msg = []
    for k in ("n_nodes", "bvh_depth", "n_triangles", "stack_entries"):
        if sh[k] != sd[k]:
            ok = False
            msg.append(f"{k}: host {sh[k]} device {sd[k]}")
    if nh.shape == nd.shape:
        boxes = np.array_equal(nh[:, :12], nd[:, :12])
        codes = np.array_equal(nh[:, 12:].view(np.int32), nd[:, 12:].view(np.int32))
        if not boxes:
            bad = np.nonzero((nh[:, :12] != nd[:, :12]).any(axis=1))[0]
            msg.append(f"{bad.size} nodes differ in their boxes, first {bad[:5]}")
        if not codes:
            bad = np.nonzero((nh[:, 12:].view(np.int32) != nd[:, 12:].view(np.int32)).any(axis=1))[0]
            msg.append(f"{bad.size} nodes differ in their child codes, first {bad[:5]}")
        ok = ok and boxes and codes
    if th.shape == td.shape:
        same = np.array_equal(th.view(np.uint32), td.view(np.uint32))
        if not same:
            bad = np.nonzero((th.view(np.uint32) != td.view(np.uint32)).any(axis=1))[0]
            msg.append(f"{bad.size} triangle records differ, first {bad[:5]}")
        ok = ok and same
    if not np.array_equal(ih, idd):
        ok = False
        msg.append(f"{int((ih != idd).sum())} primary hits differ")
    rel = abs(sh["sah_cost"] - sd["sah_cost"]) / max(1e-30, abs(sh["sah_cost"]))
    if rel > 1e-9:
        ok = False
        msg.append(f"SAH cost host {sh['sah_cost']} device {sd['sah_cost']}")
    print(f"{name}: {sh['n_triangles']} triangles, {sh['n_nodes']} nodes, depth {sh['bvh_depth']}; "
          f"build host {sh['bvh_build_ms']:.1f} ms device {sd['bvh_build_ms']:.1f} ms "
          f"(create {dth:.2f} s / {dtd:.2f} s): {'SAME TREE' if ok else 'DIFFERENT: ' + '; '.join(msg)}", flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", nargs="*", default=["Box", "Veggie"])
    ap.add_argument("--lattice", nargs="*", type=int, default=[3])
    args = ap.parse_args()
    ok = True
    for name in args.scenes:
        scene = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"))
        ok = compare(name, scene) and ok
    veggie = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "Veggie.npz"))
    for n in args.lattice:
        ok = compare(f"Veggie lattice {n}", veggie.instance_lattice(10, n)) and ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
