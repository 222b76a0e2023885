"""A whole render from host arrays to a host image: scene upload and tree build (pt_create), 256 samples per
pixel at 1080p, read-back of the float sum over PCIe -- the rate that includes what bench.py leaves out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
for name in ("Box", "Veggie"):
    s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes", name + ".npz"))
    for rep in range(3):
        t0 = time.perf_counter()
        r = pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=8)
        t1 = time.perf_counter()
        r.draw(256)
        r.synchronize()
        t2 = time.perf_counter()
        img = r.read_sum()
        t3 = time.perf_counter()
        r.close()
    paths = 1920 * 1080 * 256
    print(f"{name}: create {1e3*(t1-t0):.1f} ms, render {1e3*(t2-t1):.1f} ms ({paths/(t2-t1)/1e6:.0f} Mpaths/s), "
          f"read-back of {img.nbytes/1e6:.0f} MB {1e3*(t3-t2):.1f} ms; host to host {paths/(t3-t0)/1e6:.0f} Mpaths/s")
