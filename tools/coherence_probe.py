#!/usr/bin/env python3
"""How much of the traversal cost of scattered rays is their order?  Traces the same
secondary-like rays (origins on the visible surfaces, uniformly random directions) in
several orders; run under `rocprofv3 --kernel-trace` and read the durations of the
k_extend_bvh<false, false, L, false, false> launches in sequence (tools/coherence_probe.py
prints the order of the cases)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402

scene_name = sys.argv[1] if len(sys.argv) > 1 else "Box"
scene = pt.Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", scene_name + ".npz"))
if len(sys.argv) > 2:  # instanced on an N^3 lattice
    scene = scene.instance_lattice(10, int(sys.argv[2]))
W, H = 1920, 1080
r = pt.Renderer(scene, pt.Camera(resolution=(W, H)), max_depth=8, iterations_per_batch=4)
r.set_exact_readbacks(False)
ids, t, dirs = r.read_hit_ids(0, True, True)
hit = (ids >= 0).reshape(-1)
eye = np.array([0.0, 1.5, 15.0], np.float32)
pts = (eye + dirs.reshape(-1, 3) * t.reshape(-1, 1))[hit]
rng = np.random.default_rng(1)
rep = 4
o = np.tile(pts, (rep, 1)).astype(np.float32)  # pixel order, repeated
d = rng.normal(size=o.shape).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
o += d * 1e-3
n = o.shape[0]


def morton3(q):  # q: n x 3 ints < 1024
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


lo, hi = o.min(0), o.max(0)
cell = np.clip(((o - lo) / (hi - lo + 1e-6) * 1024).astype(np.int64), 0, 1023)
m_o = morton3(cell)
octant = (d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2)
dcell = np.clip(((d + 1) * 0.5 * 16).astype(np.int64), 0, 15)
m_d = morton3(dcell)  # 12 bits of direction
orders = {
    "pixel order (origins coherent, directions random)": np.arange(n),
    "shuffled": rng.permutation(n),
    "octant, then origin": np.lexsort((m_o, octant)),
    "direction cell (4096), then origin": np.lexsort((m_o, m_d)),
    "origin cell (32^3), then direction cell": np.lexsort((m_d, m_o >> np.uint64(15))),
    "within blocks of 2048 of pixel order: by direction cell": None,
}
blk = np.arange(n) // 2048
orders["within blocks of 2048 of pixel order: by direction cell"] = np.lexsort((m_d, blk))
ref = None
for name, idx in orders.items():
    tt, ii = r.trace_rays(o[idx], d[idx])
    back = np.empty(n, np.int64)
    back[idx] = np.arange(n)
    if ref is None:
        ref = ii[back]
    assert np.array_equal(ii[back], ref)
    print(f"{n} rays: {name}", flush=True)
r.close()
