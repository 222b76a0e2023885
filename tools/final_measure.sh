#!/bin/bash
# The measurements the documents quote, in one go on the GPU box: profiles of Box and Veggie, the default bench line,
# every configuration.  usage: tools/final_measure.sh TAG
T=$1
bash tools/profile_gpu.sh ${T}_box > gpurun_out/prof_${T}_box.log 2>&1 || exit 1
bash tools/profile_gpu.sh ${T}_veggie --scene Veggie > gpurun_out/prof_${T}_veggie.log 2>&1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/bench_${T}.json 2> gpurun_out/bench_${T}.err || exit 1
timeout -k 10 300 python tools/bench_configs.py --out gpurun_out/${T}_all_configs.json > gpurun_out/${T}_all_configs.log 2>&1 || exit 1
head -8 gpurun_out/prof/${T}_box/summary.txt
python - <<PY
import json
for t in ("kt", "kt_alone"):
    d = json.loads(open("gpurun_out/prof/${T}_box/%s.json" % t).read().strip().splitlines()[-1]); print(t, round(d["value"]))
PY
