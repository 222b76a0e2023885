import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(env_extra, args, label):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", *args], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = d["roofline"]["kernels"]
        print(f"{label}: {d['value']:.0f} Mpaths/s  extend {k['extend']['ms_total']/k['extend']['launches']*1e3:.0f} us  shade {k['shade']['ms_total']/k['shade']['launches']*1e3:.0f} us /launch; gates {d['gates'].get('primary_hit_ids_bit_exact_vs_reference')}", flush=True)
    except Exception:
        # A failed GPU run ends the whole sweep: nothing more is started on the card.
        print("failed", label, out.stderr[-800:], flush=True)
        sys.exit(1)
if __name__ == "__main__":
    for scene in ("Box", "Veggie"):
        for lds in ("0", "1"):
            for spp in ("4", "16"):
                run({"PT_SCENE_LDS": lds}, ["--scene", scene, "--spp-per-step", spp, "--steps", "8", "--warmup", "2"], f"{scene} lds={lds} spp/step={spp}")
