#!/bin/bash
# Throughput against the batch size: tools/batch_spp_ab.sh "SCENE WIDTH HEIGHT DEPTH SPP_PER_STEP" "BATCHES..." (two rounds)
CFG=${1:-"Veggie 1920 1080 8 576"}; BATCHES=${2:-"32 48 64"}
read scene w h d spp <<< "$CFG"
for rep in 1 2; do for b in $BATCHES; do
  python bench.py --scene $scene --width $w --height $h --depth $d --batch-spp $b --spp-per-step $spp --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-alone > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); print(*sys.argv[1:], round(d['value']))" $scene ${w}x$h d$d batch $b
done; done
