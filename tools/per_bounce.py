#!/usr/bin/env python3
"""Per-bounce launch times of the extend and shade kernels from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra
    python tools/per_bounce.py DIR [depth]
"""
import csv
import glob
import sys

d = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_raygen" in n:
        seq.append({"extend": [], "shade": [], "raygen": us})
    elif seq and ("k_extend_bvh<" in n and ", false, false, " in n):
        seq[-1]["extend"].append(us)
    elif seq and "k_shade" in n and "records" not in n:
        seq[-1]["shade"].append(us)
full = [s for s in seq if len(s["extend"]) == depth]
for kind in ("extend", "shade"):
    m = len(full[0][kind])
    avg = [sum(s[kind][i] for s in full) / len(full) for i in range(m)]
    print(kind, " ".join(f"{a:6.0f}" for a in avg), " sum", f"{sum(avg):.0f}", "us over", len(full), "batches")
