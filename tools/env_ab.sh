#!/bin/bash
# A/B of environment knobs of a tuning build: tools/env_ab.sh LIB SCENE "A=1 B=2" "" "C=3" ...  (two rounds each)
LIB=$1; SCENE=$2; shift 2
for rep in 1 2; do for envs in "$@"; do
  env $envs PT_AMD_LIB=$PWD/$LIB python bench.py --scene $SCENE --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']; print(sys.argv[1], '[%s]' % sys.argv[2], round(d['value']), d['gates'].get('primary_hit_ids_bit_exact_vs_reference'), {n:round(a[n]['avg_launch_ms'],3) for n in a if a[n]['avg_launch_ms']})" $SCENE "$envs"
done; done
