import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_quick import run
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "pathtracer_amd/variants/*.so")))
for lib in libs:
    env = {"PT_AMD_LIB": lib} if lib else {}
    for scene in ("Box", "Veggie"):
        run(env, ["--scene", scene, "--spp-per-step", "16", "--steps", "8", "--warmup", "2"], f"{os.path.basename(lib) if lib else 'default'} {scene}")
