import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, "tools")
from bench_quick import run
for spp, steps in (("8", "32"), ("16", "16"), ("24", "12"), ("32", "8")):
    for scene in ("Box", "Veggie"):
        run({}, ["--scene", scene, "--spp-per-step", spp, "--steps", steps, "--warmup", "2"], f"spp/step {spp} {scene}")
