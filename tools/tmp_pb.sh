export TMPDIR=/tmp
for v in n80 n112; do
  export PT_AMD_LIB=$PWD/pathtracer_amd/variants/$v.so
  rm -rf /tmp/pb_$v; timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb_$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-alone --batch-streams 1 > /tmp/pb_$v.json 2>/tmp/pb_$v.err || { tail -3 /tmp/pb_$v.err; exit 1; }
  echo "== $v"; python tools/per_bounce.py /tmp/pb_$v 8
done
