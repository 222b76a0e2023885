"""How far apart the waves of a traversal block finish: sum over waves of their busy time against the sum over blocks
of the block's longest wave (x waves per block).  Needs a tuning build with -DPT_DIAG_BLOCK, which redefines
extend diagnostics word [7] as that second sum:
    tools/build_variant.sh diagblock "-DPT_DIAG_BLOCK"
    PT_AMD_LIB=$PWD/pathtracer_amd/variants/diagblock.so python tools/wave_balance.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
from conftest import scene_arrays
for name, waves in (("Veggie", 16), ("Box_Transmission", 16), ("Box", 4)):
    for depth in (1, 2, 4, 8):
        r = pt.Renderer(pt.Scene.from_arrays(**scene_arrays(name)), pt.Camera(resolution=(1920, 1080)), max_depth=depth, profile=3)
        r.draw(32)
        st = r.stats(); d = st["extend_diag"]
        print(name, "depth", depth, "launches", st["launches_extend"], "sum wave busy", d[6], "sum block max", d[7],
              "mean/max within block %.3f" % (d[6] / (waves * max(1, d[7]))), flush=True)
        r.close()
