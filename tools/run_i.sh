python tools/bench_latency.py > gpurun_out/r04i_latency_w4.json 2> gpurun_out/r04i.err
for w in 8 16; do SEMADB_AMD_LIB=$PWD/build/w$w/libsemadb_amd.so python tools/bench_latency.py > gpurun_out/r04i_latency_w$w.json 2>> gpurun_out/r04i.err; done
SEMADB_AMD_LIB=$PWD/build/w16/libsemadb_amd.so python -m pytest tests/test_gpu_search.py -m gpu -q -x --timeout 600 > gpurun_out/r04i_w16_tests.log 2>&1
python tools/bench_filter.py > gpurun_out/r04i_filter_device.json 2>> gpurun_out/r04i.err
