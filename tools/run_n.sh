SEMADB_AMD_LIB=$PWD/build/pull256/libsemadb_amd.so python tools/bench_latency.py > gpurun_out/r04n_latency_pull256.json 2> gpurun_out/r04n.err
python -m pytest tests/test_gpu_pq.py -m gpu -q -x --timeout 1500 -k "multi_wave or parity" > gpurun_out/r04n_pq_tests.log 2>&1
