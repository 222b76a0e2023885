"""Frame rate of one-sample-per-pixel draws (the reference's interactive mode: one draw() per frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes/Box.npz"))
for w, h in [(1280, 720), (1920, 1080), (512, 512)]:
    for per_call in (1, 4, 16):
        r = pt.Renderer(s, pt.Camera(resolution=(w, h)), max_depth=10, iterations_per_batch=per_call)
        r.draw(per_call); r.synchronize(); r.reset(); r.synchronize()
        n = max(4, 256 // per_call)
        t0 = time.perf_counter()
        for _ in range(n):
            r.draw(per_call)
        r.synchronize()
        dt = time.perf_counter() - t0
        r.close()
        print(f"{w}x{h} {per_call} spp per draw: {dt/n*1e3:.3f} ms per draw, {w*h*per_call*n/dt/1e6:.0f} Mpaths/s", flush=True)
