"""Scheduling figures of the traversal kernel for one scene: loop iterations and lane use per 64 rays.
usage: python tools/extend_diag.py [scene] [width height spp depth]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import pathtracer_amd as pt  # noqa: E402
from conftest import scene_arrays  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Box"
w, h, spp, depth = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 32, 8)
r = pt.Renderer(pt.Scene.from_arrays(**scene_arrays(name)), pt.Camera(resolution=(w, h)), max_depth=depth, profile=3)
r.draw(spp)
st = r.stats()
d = st["extend_diag"]
rays = st["segments"]
w64 = rays / 64.0
print(f"{name}: rays {rays}, per 64 rays: node steps {d[0] / w64:.2f} (lanes {d[1] / max(1, d[0]):.1f}), "
      f"triangle steps {d[2] / w64:.2f} (lanes {d[3] / max(1, d[2]):.1f}), rounds {d[4] / w64:.2f} "
      f"(lanes holding a ray {d[5] / max(1, d[4]):.1f}); nodes/ray {st['nodes_visited'] / rays:.2f}, "
      f"tris/ray {st['tris_tested'] / rays:.2f}")
r.close()
