import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
from conftest import scene_arrays, hit_fixture
for name in ["sphere", "metal_rough_smooth", "Box_Metallic"]:
    fx = hit_fixture(name)["ids"]
    for bvh in (True, False):
        r = pt.Renderer(pt.Scene.from_arrays(**scene_arrays(name)), pt.Camera(resolution=(256, 256)), use_bvh=bvh)
        ids, t, dirs = r.read_hit_ids(0, True, True)
        r.close()
        bad = np.argwhere(ids != fx)
        print(name, "bvh" if bvh else "scan", "mismatches", len(bad))
        for y, x in bad[:5]:
            print("  px", y, x, "gpu", ids[y, x], "ref", fx[y, x], "t", t[y, x], t[y, x].view(np.uint32), "dir", dirs[y, x], [hex(v) for v in dirs[y, x].view(np.uint32)])
