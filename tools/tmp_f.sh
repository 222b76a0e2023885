for sc in Box_Transmission Box_TransmissionSphere Box_Anisotropic AnisotropicTest; do for fm in 0 1000000000 0 1000000000; do
  PT_FUSED_MAX=$fm PT_AMD_LIB=$PWD/pathtracer_amd/variants/b5.so python bench.py --scene $sc --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-alone > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], 'fused_max', sys.argv[2], round(d['value']))" $sc $fm
done; done
PT_FUSED_MAX=1000000000 PT_AMD_LIB=$PWD/pathtracer_amd/variants/b5.so python bench.py --scene Veggie --lattice 19 --width 3840 --height 2160 --spp-per-step 32 --batch-spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-alone | python -c "import json,sys; print('lattice fused', round(json.load(sys.stdin)['value']))"
PT_FUSED_MAX=0 PT_AMD_LIB=$PWD/pathtracer_amd/variants/b5.so python bench.py --scene Veggie --lattice 19 --width 3840 --height 2160 --spp-per-step 32 --batch-spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-alone | python -c "import json,sys; print('lattice separate', round(json.load(sys.stdin)['value']))"
