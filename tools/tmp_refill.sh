export PT_AMD_LIB=$PWD/pathtracer_amd/variants/cur.so
for rf in 16 8 12 24 32; do for sc in Box Veggie; do
  PT_REFILL_AT=$rf python bench.py --scene $sc --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']['kernels']; print('refill', sys.argv[1], sys.argv[2], round(d['value']), {n:round(a[n]['avg_launch_ms'],3) for n in ('extend','shade')})" $rf $sc
done; done
