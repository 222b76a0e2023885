run() { python bench.py --scene $SC --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']['kernels']; print(sys.argv[1], round(d['value']), {n:round(a[n]['avg_launch_ms'],3) for n in ('extend','shade')})" "$SC $1"; }
for SC in Box Veggie; do
export PT_AMD_LIB=$PWD/pathtracer_amd/variants/tune.so
run "default"
PT_EXT_PER_CU=4 run "extend capped at 4 blocks"
PT_EXT_PER_CU=3 run "extend capped at 3 blocks"
export PT_AMD_LIB=$PWD/pathtracer_amd/variants/sw5.so
run "shade 5 waves/SIMD"
PT_EXT_PER_CU=4 run "shade 5 waves/SIMD, extend capped at 4"
done
