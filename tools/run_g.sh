python -m pytest tests/test_gpu_filter.py tests/test_gpu_versions.py tests/test_gpu_delete.py -m gpu -q --timeout 900 -x > gpurun_out/r04g_filter.log 2>&1
python tools/bench_filter.py > gpurun_out/r04g_filter_device.json 2> gpurun_out/r04g_filter.err
for w in 2 8; do SEMADB_AMD_LIB=$PWD/build/w$w/libsemadb_amd.so python tools/bench_latency.py > gpurun_out/r04g_latency_w$w.json 2>> gpurun_out/r04g_filter.err; done
