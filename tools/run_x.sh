timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 > gpurun_out/r04ad_full_gpu_tests.log 2>&1; echo "tests rc $?" > gpurun_out/r04ad_rc.log
for seed in 411 412; do timeout 400 python tools/fuzz_parity.py --trials 200 --seed $seed > gpurun_out/r04ad_fuzz_$seed.json 2> gpurun_out/r04ad_fuzz_$seed.err; echo "seed $seed rc $?" >> gpurun_out/r04ad_rc.log; done
