"""Ad-hoc GPU bring-up check (not part of the test suite)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
import oracle_binding as ob

def load(name):
    return pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes", name + ".npz"))

print(pt.version(), "devices", pt.device_count(), flush=True)
for name in ["Box", "Veggie", "Box_Transmission", "Box_Anisotropic"]:
    s = load(name)
    cam = pt.Camera(resolution=(256, 256)); cam.update()
    osc = ob.OracleScene(s.arrays())
    oc = ob.make_camera(cam.eye, cam.forward, cam.up, cam.right, 256, 256)
    oid, ot, od = ob.primary(osc, oc, 0)
    for bvh in (True, False):
        r = pt.Renderer(s, cam, max_depth=8, use_bvh=bvh)
        gid, gt, gd = r.read_hit_ids(0, True, True)
        print(name, "bvh" if bvh else "scan", "id mismatches", int((gid != oid).sum()),
              "t bit mismatches", int((gt.view(np.uint32) != ot.view(np.uint32)).sum()),
              "dir bit mismatches", int((gd.view(np.uint32) != od.view(np.uint32)).sum()), r.stats()["bvh_depth"], flush=True)
        if bvh:
            t0 = time.time(); r.draw(64); r.synchronize(); dt = time.time() - t0
            img = r.read_radiance()
            st = r.stats()
            print("  gpu 64spp %.3fs mean" % dt, img[..., :3].mean(axis=(0, 1)), "segments/path %.3f" % (st["segments"] / st["paths"]), "nan", st["nan_samples"], flush=True)
        r.close()
    t0 = time.time()
    osum, ost = ob.render(osc, oc, 0, 16, 8, nthreads=16)
    print("  oracle 16spp %.2fs mean" % (time.time() - t0), osum[..., :3].mean(axis=(0, 1)) / 16, ost, flush=True)

# throughput at 1080p
for name in ["Box", "Veggie"]:
    s = load(name)
    cam = pt.Camera(resolution=(1920, 1080))
    r = pt.Renderer(s, cam, max_depth=8)
    r.draw(8); r.synchronize()
    t0 = time.time(); r.draw(64); r.synchronize(); dt = time.time() - t0
    print(name, "1080p depth 8: %.1f Mpaths/s" % (1920 * 1080 * 64 / dt / 1e6), flush=True)
    r.close()
    r = pt.Renderer(s, cam, max_depth=8, profile=True)
    r.draw(8); r.synchronize()
    print("  profile:", {k: v for k, v in r.stats().items() if k.startswith(("ms_", "launches", "nodes", "tris", "segments", "paths"))}, flush=True)
    r.close()
