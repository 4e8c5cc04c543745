export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > gpurun_out/r01h_bench.log 2>gpurun_out/r01h_bench.err
tail -1 gpurun_out/r01h_bench.log | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01h_trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r01h_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01h_pmc_fetch -- python3 tools/pmc_run.py > gpurun_out/r01h_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r01h_pmc_write -- python3 tools/pmc_run.py > gpurun_out/r01h_pmc_write.log 2>&1
# keep only the rows the summaries use (the build launches tens of thousands of small kernels)
for f in $(find gpurun_out -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
  head -1 $f > $f.tmp; grep -E "k_index_distance|k_greedy_search" $f >> $f.tmp; mv $f.tmp $f
done
python tools/gather_ceiling.py > gpurun_out/r01h_gather.json 2>/dev/null
python tools/bench_pq.py 2>&1 | tail -1 > gpurun_out/r01h_c4_1M.log
du -sh gpurun_out
