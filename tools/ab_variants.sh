#!/bin/bash
# A/B of library variants in one box: tools/ab_variants.sh [bench args] -- lib1.so lib2.so ...   ("default" = the product build)
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = default ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$PWD/$lib; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "${ARGS[@]}" > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']; print(sys.argv[1], round(d['value']), d['gates'].get('primary_hit_ids_bit_exact_vs_reference'), {n:round(a[n]['avg_launch_ms'],3) for n in a})" "$lib"
done; done
