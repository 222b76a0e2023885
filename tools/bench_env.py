"""Runs bench.py under several environment settings (tuning experiments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_quick import run
# usage: bench_env.py "A=1 B=2" "A=3" ...   ("-" = defaults)
for spec in sys.argv[1:] or ["-"]:
    env = {} if spec == "-" else dict(kv.split("=", 1) for kv in spec.split())
    for scene in ("Box", "Veggie"):
        run(env, ["--scene", scene, "--steps", "8", "--warmup", "2"], f"[{spec}] {scene}")
