# usage (GPU box, repo root): bash tools/collect_profiles.sh <tag>
# bench log, rocprofv3 kernel trace of the same command, the two PMC passes of the guide's HBM section, the gather
# ceiling.  tools/make_profile_summary.py <tag> gpurun_out/ gpurun_out/<tag>_bench.log turns them into profiles/.
tag=${1:-r02}
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.log 2>gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.log | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-host-rates > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 tools/pmc_run.py gpurun_out/pmc_expected.json > gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 tools/pmc_run.py gpurun_out/pmc_expected.json > gpurun_out/${tag}_pmc_write.log 2>&1
# keep only the rows the summaries use (the build launches tens of thousands of small kernels)
for f in $(find gpurun_out -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
  head -1 $f > $f.tmp; grep -E "k_index_distance|k_greedy_search" $f >> $f.tmp; mv $f.tmp $f
done
python tools/gather_ceiling.py > gpurun_out/${tag}_gather.json 2>/dev/null
du -sh gpurun_out
