"""Condenses the rocprofv3 CSVs written by tools/profile_gpu.sh into one text
summary: per-kernel time (kernel trace) and per-kernel counter sums (PMC
passes).  FETCH_SIZE is doubled for gfx950 (it tallies 128-byte requests as
64 bytes on wide streaming reads); both raw and corrected values are shown."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
machine = {}  # kernel -> figures per launch, written to <out>/pmc.json


def short(name):
    m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


# kt: the run as benchmarked (two batches in flight, launches of the two streams overlap);
# kt_alone: the same run with one batch at a time, every kernel alone on the GPU -- the
# duration the counters of the PMC passes (which serialise kernels) belong to.
for sub, title, key in (("kt", "kernel trace (rocprofv3 --kernel-trace --stats), two batches in flight", "avg_launch_us_overlapped"),
                        ("kt_alone", "kernel trace, one batch at a time (--batch-streams 1)", "avg_launch_us")):
    files = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        continue
    print(f"== {title} ==")
    for f in files:
        for row in csv.DictReader(open(f)):
            print(f"{short(row['Name']):34s} calls {int(row['Calls']):5d}  total {float(row['TotalDurationNs'])/1e6:9.3f} ms  "
                  f"avg {float(row['AverageNs'])/1e3:9.2f} us  {float(row['Percentage']):6.2f} %")
            machine.setdefault(short(row['Name']), {})[key] = float(row['AverageNs']) / 1e3
            machine[short(row['Name'])]["launches"] = int(row['Calls'])
    print()
for m in machine.values():  # a profile without the second trace: the only duration there is
    m.setdefault("avg_launch_us", m.get("avg_launch_us_overlapped", 0.0))

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
                machine.setdefault(k, {})["hbm_read_bytes_per_launch"] = 2 * v * 1024 / n
                parts.append(f"FETCH_SIZE {v/1024/1024:.2f} GiB raw (x2 = {2*v/1024/1024:.2f} GiB corrected; unit KiB) over {n} dispatches")
            elif c == "WRITE_SIZE":
                machine.setdefault(k, {})["hbm_write_bytes_per_launch"] = v * 1024 / n
                parts.append(f"WRITE_SIZE {v/1024/1024:.2f} GiB (unit KiB) over {n} dispatches")
            else:
                parts.append(f"{c} {v:.4g}")
        print(f"{k:34s} " + "; ".join(parts))
    for k in sorted(agg):
        a = agg[k]
        if "TCC_HIT_sum" in a:
            print(f"   {k:31s} L2 hit rate {a['TCC_HIT_sum']/max(1.0, a['TCC_HIT_sum']+a['TCC_MISS_sum']):.4f}")
        if "TCC_HIT_sum" in a:
            machine.setdefault(k, {})["l2_hit_rate"] = a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
        if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
            machine.setdefault(k, {})["active_lanes_per_valu_instruction"] = a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"]
            machine[k]["valu_instructions_per_launch"] = a.get("SQ_INSTS_VALU", 0.0) / max(1, calls[(k, "SQ_INSTS_VALU")])
            # SQ_ACTIVE_INST_VALU counts 4-cycle units summed over all SIMDs
            machine[k]["valu_busy_quads_per_launch"] = a["SQ_ACTIVE_INST_VALU"] / max(1, calls[(k, "SQ_ACTIVE_INST_VALU")])
            print(f"   {k:31s} active lanes per VALU instruction {a['SQ_THREAD_CYCLES_VALU']/a['SQ_ACTIVE_INST_VALU']:.1f} of 64; "
                  f"VALU-active share of wave cycles {a['SQ_ACTIVE_INST_VALU']/max(1.0, a['SQ_WAVE_CYCLES']):.3f}; "
                  f"busy cycles {a.get('SQ_BUSY_CYCLES', 0):.4g}")

# VALU issue utilisation: busy 4-cycle units / (launch duration x 2.4 GHz / 4 x 1024 SIMDs)
for k, m in machine.items():
    if "valu_busy_quads_per_launch" in m and "avg_launch_us" in m:
        m["valu_issue_utilisation"] = m["valu_busy_quads_per_launch"] / (m["avg_launch_us"] * 1e-6 * 2.4e9 / 4 * 1024)
    if "hbm_read_bytes_per_launch" in m and "avg_launch_us" in m:
        m["hbm_GBs"] = (m["hbm_read_bytes_per_launch"] + m.get("hbm_write_bytes_per_launch", 0.0)) / (m["avg_launch_us"] * 1e-6) / 1e9
with open(os.path.join(out, "pmc.json"), "w") as f:
    json.dump({k: v for k, v in machine.items() if k.startswith("k_")}, f, indent=1, sort_keys=True)
print("\n== per launch (pmc.json) ==")
for k, m in sorted(machine.items()):
    if k.startswith("k_"):
        print(f"{k:34s} " + "; ".join(f"{a} {b:.4g}" for a, b in sorted(m.items())))
