python -m pytest tests/test_gpu_filter.py tests/test_gpu_versions.py -m gpu -q -x --timeout 900 > gpurun_out/r04r_filter_tests.log 2>&1
for seed in 401 402 403; do timeout 600 python tools/fuzz_parity.py --trials 250 --seed $seed > gpurun_out/r04r_fuzz_$seed.json 2> gpurun_out/r04r_fuzz_$seed.err; echo "seed $seed rc $?" >> gpurun_out/r04r_fuzz_rc.log; done
