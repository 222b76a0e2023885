"""Load balance of the extend kernel's persistent waves: per launch, mean and longest wave busy time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
for scene in sys.argv[1:] or ("Box", "Veggie"):
    s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes", scene + ".npz"))
    for depth in (1, 2):
        r = pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=depth, iterations_per_batch=16, profile=3)
        r.draw(16)
        st = r.stats(); r.close()
        d = st["extend_diag"]
        print(f"{scene} depth {depth}: launches {st['launches_extend']}, extend {st['ms_extend']*1e3:.0f} us total; "
              f"sum of wave busy {d[6]/100:.0f} us (/4096 = {d[6]/100/4096:.0f} us), longest wave {d[7]/100:.0f} us", flush=True)
