python tools/bench_latency.py > gpurun_out/r04m_latency_pull.json 2> gpurun_out/r04m.err
SEMADB_AMD_LIB=$PWD/build/nopull/libsemadb_amd.so python tools/bench_latency.py > gpurun_out/r04m_latency_nopull.json 2>> gpurun_out/r04m.err
python -m pytest tests/test_gpu_search.py -m gpu -q -x --timeout 600 > gpurun_out/r04m_tests.log 2>&1
