mkdir -p gpurun_out/r3b
SEL="converged_image_against_reference or image_against_reference_at_1920x1080 or rgba8_running_average or non_finite_samples_are_counted or sample_surface_matches_oracle or short_render_agrees"
for v in fast fast_ds; do
  PT_AMD_LIB=$PWD/pathtracer_amd/variants/$v.so timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q -s -k "$SEL" --timeout 280 --timeout-method thread > gpurun_out/r3b/pytest_$v.log 2>&1
  echo "$v rc=$?"; grep -E "passed|failed|relMSE|BSDF samples|FAILED" gpurun_out/r3b/pytest_$v.log | tail -12
done
timeout -k 10 600 tools/ab_variants.sh -- pathtracer_amd/variants/base.so pathtracer_amd/variants/fast.so pathtracer_amd/variants/fast_ds.so > gpurun_out/r3b/ab_box.log 2>&1; cat gpurun_out/r3b/ab_box.log
timeout -k 10 600 tools/ab_variants.sh --scene Veggie -- pathtracer_amd/variants/base.so pathtracer_amd/variants/fast.so pathtracer_amd/variants/fast_ds.so > gpurun_out/r3b/ab_veggie.log 2>&1; cat gpurun_out/r3b/ab_veggie.log
