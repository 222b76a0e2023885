import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_quick import run
for scene in ("Box", "Veggie"):
    for shift in ("8", "10"):
        run({"PT_REGION_SHIFT": shift}, ["--scene", scene, "--spp-per-step", "16", "--steps", "8", "--warmup", "2"], f"{scene} region_shift={shift}")
run({}, ["--scene", "Box", "--spp-per-step", "4", "--steps", "16", "--warmup", "2"], "Box spp/step=4")
