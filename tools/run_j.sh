python -m pytest tests/test_gpu_search.py tests/test_gpu_nonfinite.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 1500 -k "not c3_build and not c4" > gpurun_out/r04j_tests.log 2>&1
python tools/bench_latency.py > gpurun_out/r04j_latency.json 2> gpurun_out/r04j.err
