import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import pathtracer_amd as pt
s = pt.Scene.load_npz("tests/golden/scenes/Box.npz")
t0 = time.perf_counter()
r = pt.Renderer(s, pt.Camera(resolution=(3840, 2160)), max_depth=4, iterations_per_batch=128)
print("created in", round(time.perf_counter() - t0, 2), "s", flush=True)
r.draw(3); r.draw(2)
a = r.read_sum(); r.close()
r = pt.Renderer(s, pt.Camera(resolution=(3840, 2160)), max_depth=4, iterations_per_batch=1)
r.draw(5); b = r.read_sum(); r.close()
print("same image:", np.array_equal(a, b))
