import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
for name in ("Box", "Veggie"):
    s = pt.Scene.load_npz(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + f"/tests/golden/scenes/{name}.npz")
    prev = 0
    out = []
    for d in range(1, 9):
        r = pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=d, iterations_per_batch=4)
        r.draw(4); r.synchronize()
        st = r.stats(); r.close()
        seg = st["segments"] / st["paths"]
        out.append(seg - prev); prev = seg
    print(name, " ".join(f"{x:.3f}" for x in out))
