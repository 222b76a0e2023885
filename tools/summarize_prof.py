"""Condenses the rocprofv3 CSVs written by tools/profile_gpu.sh into one text
summary: per-kernel time (kernel trace) and per-kernel counter sums (PMC
passes).  FETCH_SIZE is doubled for gfx950 (it tallies 128-byte requests as
64 bytes on wide streaming reads); both raw and corrected values are shown."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print(f"{short(row['Name']):34s} calls {int(row['Calls']):5d}  total {float(row['TotalDurationNs'])/1e6:9.3f} ms  "
              f"avg {float(row['AverageNs'])/1e3:9.2f} us  {float(row['Percentage']):6.2f} %")

for p in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(p):
        continue
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(k, row["Counter_Name"])] += 1
    print(f"\n== {os.path.basename(p)} (sums over all dispatches) ==")
    for k in sorted(agg):
        if not k.startswith("k_"):
            continue
        parts = []
        for c, v in sorted(agg[k].items()):
            n = calls[(k, c)]
            if c == "FETCH_SIZE":
                parts.append(f"FETCH_SIZE {v/1024/1024:.2f} GiB raw (x2 = {2*v/1024/1024:.2f} GiB corrected; unit KiB) over {n} dispatches")
            elif c == "WRITE_SIZE":
                parts.append(f"WRITE_SIZE {v/1024/1024:.2f} GiB (unit KiB) over {n} dispatches")
            else:
                parts.append(f"{c} {v:.4g}")
        print(f"{k:34s} " + "; ".join(parts))
    for k in sorted(agg):
        a = agg[k]
        if "TCC_HIT_sum" in a:
            print(f"   {k:31s} L2 hit rate {a['TCC_HIT_sum']/max(1.0, a['TCC_HIT_sum']+a['TCC_MISS_sum']):.4f}")
        if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
            print(f"   {k:31s} active lanes per VALU instruction {a['SQ_THREAD_CYCLES_VALU']/a['SQ_ACTIVE_INST_VALU']:.1f} of 64; "
                  f"VALU-active share of wave cycles {a['SQ_ACTIVE_INST_VALU']/max(1.0, a['SQ_WAVE_CYCLES']):.3f}; "
                  f"busy cycles {a.get('SQ_BUSY_CYCLES', 0):.4g}")
