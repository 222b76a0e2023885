for sc in Box Veggie Box_Transmission; do for v in n80 n112; do for kb in 40 32 27; do
  PT_LDS_BUDGET_KB=$kb PT_AMD_LIB=$PWD/pathtracer_amd/variants/$v.so python bench.py --scene $sc --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']['kernels']; print(sys.argv[1], sys.argv[2], 'budget', sys.argv[3], round(d['value']), {n:round(a[n]['avg_launch_ms'],3) for n in ('extend',)})" $sc $v $kb
done; done; done
