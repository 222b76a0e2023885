"""Frame time of one-sample-per-pixel draws (the reference's interactive mode: one draw() per frame,
renderer.cpp:124-150): throughput of back-to-back draws, and latency of draw + synchronize."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes/Box.npz"))
for w, h in [(1280, 720), (1920, 1080), (512, 512)]:
    for per_call in (1, 4):
        r = pt.Renderer(s, pt.Camera(resolution=(w, h)), max_depth=10)
        for _ in range(6):
            r.draw(per_call)
        r.synchronize(); r.reset(); r.synchronize()
        n = 256
        t0 = time.perf_counter()
        for _ in range(n):
            r.draw(per_call)
        r.synchronize()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(n):
            r.draw(per_call)
            r.synchronize()
        dl = time.perf_counter() - t0
        r.close()
        print(f"{w}x{h} {per_call} spp per draw: {dt/n*1e3:.3f} ms per draw back to back ({w*h*per_call*n/dt/1e6:.0f} Mpaths/s), "
              f"{dl/n*1e3:.3f} ms per draw + synchronize", flush=True)
