run() { python bench.py --scene Veggie --lattice 19 --width 3840 --height 2160 --spp-per-step 32 --batch-spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); a=d['roofline']['alone']['kernels']; print(sys.argv[1], round(d['value']), {n:round(a[n]['avg_launch_ms'],3) for n in ('extend','shade')})" "$1"; }
export PT_AMD_LIB=$PWD/pathtracer_amd/variants/w8.so
PT_DEEP_STACK=12 PT_DEEP_BUDGET_KB=20 run "w8: 12 entries, 20 KB"
PT_DEEP_STACK=16 PT_DEEP_BUDGET_KB=20 run "w8: 16 entries, 20 KB"
PT_DEEP_STACK=16 PT_DEEP_BUDGET_KB=26 run "w8: 16 entries, 26 KB"
export PT_AMD_LIB=$PWD/pathtracer_amd/variants/w7.so
PT_DEEP_STACK=12 PT_DEEP_BUDGET_KB=22 run "w7: 12 entries, 22 KB"
PT_DEEP_STACK=16 PT_DEEP_BUDGET_KB=22 run "w7: 16 entries, 22 KB"
export PT_AMD_LIB=$PWD/pathtracer_amd/variants/tune.so
PT_DEEP_STACK=16 PT_DEEP_BUDGET_KB=31 run "tune: 16 entries, 31 KB"
PT_DEEP_STACK=16 PT_DEEP_BUDGET_KB=24 run "tune: 16 entries, 24 KB"
