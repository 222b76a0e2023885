import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes", "Box.npz"))
for depth in (1, 2):
    r = pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=depth, iterations_per_batch=16, profile=3)
    r.draw(16)
    st = r.stats(); r.close()
    d = st["extend_diag"]
    print(f"depth {depth}: extend {st['ms_extend']*1e3:.0f} us in {st['launches_extend']} launches; waves by busy time in 200 us bins: {d[:6]}; sum busy {d[6]/100:.0f} us; longest wave {d[7]/100:.0f} us")
