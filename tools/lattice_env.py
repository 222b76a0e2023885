"""bench.py on the 10-million-triangle scene under several environment settings (tuning experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_quick import run
for spec in sys.argv[1:] or ["-"]:
    env = {} if spec == "-" else dict(kv.split("=", 1) for kv in spec.split())
    run(env, ["--scene", "Veggie", "--lattice", "19", "--width", "3840", "--height", "2160", "--spp-per-step", "8",
              "--steps", "6", "--warmup", "2", "--no-alone"], f"[{spec}] lattice")
