python -m pytest tests/test_gpu_filter.py tests/test_gpu_host_mirror.py -m gpu -q -x --timeout 900 > gpurun_out/r04q_filter_tests.log 2>&1
python tools/bench_filter.py > gpurun_out/r04q_filter.json 2> gpurun_out/r04q.err
