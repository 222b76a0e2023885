#!/bin/bash
# A/B of the hot-triangle prefix in LDS (tuning build): tools/hot_ab.sh LIB SCENE...
LIB=$1; shift
for scene in "$@"; do for rep in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export PT_NO_HOT_TRIS=1; else unset PT_NO_HOT_TRIS; fi
  PT_AMD_LIB=$PWD/$LIB python bench.py --scene $scene --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']; print(sys.argv[1], 'no_hot=$v', round(d['value']), d['gates'].get('primary_hit_ids_bit_exact_vs_reference'), {n:round(a[n]['avg_launch_ms'],3) for n in a})" $scene
done; done; done
