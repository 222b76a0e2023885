#!/bin/bash
# One rocprofv3 counter pass over a short bench.py run, summed per kernel:
#   tools/pmc_pass.sh NAME "COUNTER COUNTER ..." [bench args...]   -> gpurun_out/pmc/NAME.txt
set -u
NAME=$1; COUNTERS=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/pmc/$NAME
mkdir -p "$OUT"
timeout -s KILL 150 rocprofv3 --pmc $COUNTERS --output-format csv -d "$OUT" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-alone "$@" > "$OUT.json" 2> "$OUT.err" || { tail -5 "$OUT.err"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k in sorted(acc, key=lambda k: -n[k])[:6]:
    print(k, "dispatches", n[k], {c: f"{v / n[k]:.4g}" for c, v in acc[k].items()})
PY
