timeout 600 python -m pytest tests/test_gpu_host_mirror.py tests/test_gpu_cluster.py -m gpu -q -x --timeout 300 > gpurun_out/r04ae_host_tests.log 2>&1; echo "tests rc $?" > gpurun_out/r04ae_rc.log
timeout 1500 bash tools/collect_profiles.sh r04b > gpurun_out/r04b_collect.log 2>&1; echo "collect rc $?" >> gpurun_out/r04ae_rc.log
