import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for shift in (8, 9, 10, 11):
    for spp in (4, 8, 16):
        env = dict(os.environ, PT_REGION_SHIFT=str(shift))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--spp-per-step", str(spp),
                              "--steps", str(max(4, 64 // spp)), "--warmup", "2"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = d["roofline"]["kernels"]
            print(f"shift {shift} spp/step {spp}: {d['value']:.0f} Mpaths/s  extend {k['extend']['ms_total']/k['extend']['launches']*1e3:.0f} us  shade {k['shade']['ms_total']/k['shade']['launches']*1e3:.0f} us (per launch)", flush=True)
        except Exception as e:
            print("failed", shift, spp, out.stderr[-500:])
