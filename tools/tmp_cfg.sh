for v in n80 n112; do
  PT_AMD_LIB=$PWD/pathtracer_amd/variants/$v.so python tools/bench_configs.py --out /tmp/c_$v.json > /tmp/c_$v.log 2>&1 || { tail -3 /tmp/c_$v.log; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/c_$v.json')); print('$v', {c['name']: (round(c['mpaths_per_s']), round(c['extend_us_per_launch'])) for c in d})"
done
