import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_extend_bvh<false, false" in r["Kernel_Name"] and "false, false>" in r["Kernel_Name"]]
print("extend launches (us):", " ".join(f"{d:.0f}" for d in ds))
