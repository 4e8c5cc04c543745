python -m pytest tests/test_gpu_pq.py -m gpu -q -x --timeout 1500 > gpurun_out/r04l_pq_tests.log 2>&1
python bench.py --config c4 --rows 1000000 --pq-m 8,128,192,384 --steps 10 > gpurun_out/r04l_c4_1M.json 2> gpurun_out/r04l_c4.err
