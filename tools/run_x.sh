timeout 300 python -m pytest tests/test_gpu_search.py tests/test_gpu_nonfinite.py tests/test_gpu_filter.py -m gpu -q -x --timeout 120 > gpurun_out/r04ac_tests.log 2>&1; echo "tests rc $?" > gpurun_out/r04ac_rc.log
timeout 300 python tools/bench_latency.py > gpurun_out/r04ac_latency_ahead64.json 2>> gpurun_out/r04ac.err; echo "lat rc $?" >> gpurun_out/r04ac_rc.log
SEMADB_AMD_LIB=$PWD/build/ahead0/libsemadb_amd.so timeout 300 python tools/bench_latency.py > gpurun_out/r04ac_latency_ahead0.json 2>> gpurun_out/r04ac.err
SEMADB_AMD_LIB=$PWD/build/ahead256/libsemadb_amd.so timeout 300 python tools/bench_latency.py > gpurun_out/r04ac_latency_ahead256.json 2>> gpurun_out/r04ac.err
