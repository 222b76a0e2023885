mkdir -p gpurun_out/r3f
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bvh_device.py tests/test_gpu_renderer.py -m gpu -x -q -k "lattice or ten_million or primary_rays_left_out" --timeout 300 --timeout-method thread > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3f/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export PT_NO_WIDE=1; else unset PT_NO_WIDE; fi
  PT_AMD_LIB=$PWD/pathtracer_amd/variants/wide.so timeout -k 10 300 python bench.py --scene Veggie --lattice 19 --width 3840 --height 2160 --steps 3 --warmup 1 --spp-per-step 64 --batch-spp 8 > gpurun_out/r3f/bench_l19_nowide$v.json 2> gpurun_out/r3f/bench_l19_nowide$v.err || tail -3 gpurun_out/r3f/bench_l19_nowide$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r3f/bench_l19_nowide$v.json')); r=d['roofline']; print('no_wide=$v', round(d['value']), {k:round(v['avg_launch_ms'],3) for k,v in r['alone'].items()}, d['setup'])"
done
