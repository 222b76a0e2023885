export PT_AMD_LIB=$PWD/pathtracer_amd/variants/tune.so
for lf in 4 3 2 6; do for sc in Box Veggie; do
  PT_BVH_LEAF=$lf python bench.py --scene $sc --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']['kernels']; print('leaf', sys.argv[1], sys.argv[2], round(d['value']), d['gates'].get('primary_hit_ids_bit_exact_vs_reference'), {n:round(a[n]['avg_launch_ms'],3) for n in ('extend','shade')}, d['setup']['bvh_nodes'], d['setup']['bvh_depth'])" $lf $sc
done; done
