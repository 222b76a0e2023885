import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt
for scene in ("Box", "Veggie"):
    s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes", scene + ".npz"))
    r = pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=8, iterations_per_batch=16, profile=2)
    r.draw(16)
    st = r.stats(); r.close()
    d = st["extend_diag"]; seg = st["segments"]
    print(scene, "rays", seg, "nodes/ray %.2f tris/ray %.2f" % (st["nodes_visited"] / seg, st["tris_tested"] / seg))
    print("  phase A: %.2f iterations per ray-wave(64 rays), lanes active %.1f/64" % (d[0] / (seg / 64), d[1] / max(1, d[0])))
    print("  phase B: %.2f tri-iterations per 64 rays, lane-tests per iteration %.1f/64" % (d[2] / (seg / 64), d[3] / max(1, d[2])))
    print("  rounds per 64 rays %.2f, lanes holding a ray %.1f/64" % (d[4] / (seg / 64), d[5] / max(1, d[4])))
    print("  wave busy time: sum %.0f us over all launches, longest single wave %.0f us" % (d[6] / 100, d[7] / 100))
    r = pt.Renderer(s, pt.Camera(resolution=(1920, 1080)), max_depth=8, iterations_per_batch=16)
    r.draw(16)
    st = r.stats(); r.close()
    print("  shade: hits %d in %d batches of 64 lanes: fill %.3f" % (st["hits_shaded"], st["shade_batches"], st["hits_shaded"] / max(1, 64 * st["shade_batches"])), flush=True)
