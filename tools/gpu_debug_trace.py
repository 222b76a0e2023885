import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt, oracle_binding as ob
from conftest import scene_arrays
from test_gpu_parity import _secondary_like_rays
for name in ["Box", "Veggie", "cube"]:
    a = scene_arrays(name)
    n = 200000 if a["indices"].shape[0] < 600 else 60000
    o, d = _secondary_like_rays(a, n, 3)
    ot, oid = ob.trace(ob.OracleScene(a), o, d)
    r = pt.Renderer(pt.Scene.from_arrays(**a), pt.Camera(resolution=(64, 64)))
    t, ids = r.trace_rays(o, d)
    r.close()
    bad = np.nonzero((ids != oid) | (t.view(np.uint32) != ot.view(np.uint32)))[0]
    k = n // 10
    print(name, "mismatch", len(bad), "axis-par", (bad < k).sum(), "vertex-aimed", ((bad >= k) & (bad < 2 * k)).sum(), "random", (bad >= 2 * k).sum(),
          "oracle t nan", np.isnan(ot[bad]).sum(), "inf", np.isinf(ot[bad]).sum())
    for i in bad[:8]:
        print("   ", i, "gpu", ids[i], t[i], "oracle", oid[i], ot[i], "o", o[i], "d", d[i])
