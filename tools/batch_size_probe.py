"""Throughput against batch size: back-to-back draws of k samples per pixel (1080p and 512^2, Box, depth 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes/Box.npz"))
for w, h in [(1920, 1080), (512, 512)]:
    for per_call in (1, 2, 4, 8, 16, 32, 64):
        if w * h * per_call > 70_000_000:
            continue
        r = pt.Renderer(s, pt.Camera(resolution=(w, h)), max_depth=8, iterations_per_batch=per_call)
        for _ in range(4):
            r.draw(per_call)
        r.synchronize(); r.reset(); r.synchronize()
        n = max(8, min(256, 512 // per_call))
        t0 = time.perf_counter()
        for _ in range(n):
            r.draw(per_call)
        r.synchronize()
        dt = time.perf_counter() - t0
        r.close()
        print(f"{w}x{h} {per_call:2d} spp per draw ({w*h*per_call/1e6:6.2f} M paths): {dt/n*1e3:8.3f} ms per draw, {w*h*per_call*n/dt/1e6:6.0f} Mpaths/s", flush=True)
