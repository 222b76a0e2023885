import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracer_amd as pt
s = pt.Scene.load_npz(ROOT + "/tests/golden/scenes/Box.npz")
def run(n_r, steps=12, spp=16):
    rs = [pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=8, iterations_per_batch=spp) for _ in range(n_r)]
    for r in rs: r.draw(spp)
    for r in rs: r.synchronize(); r.reset()
    for r in rs: r.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for r in rs: r.draw(spp)
    for r in rs: r.synchronize()
    dt = time.perf_counter() - t0
    for r in rs: r.close()
    print(f"{n_r} renderer(s): {1920*1080*spp*steps*n_r/dt/1e6:.0f} Mpaths/s", flush=True)
run(1); run(2); run(1); run(2); run(3)
