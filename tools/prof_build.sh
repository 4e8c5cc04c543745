# rocprofv3 kernel-trace summary of one full C3 build (bench.py --config c3): per-kernel time split.
# usage (on the GPU box, from the repo root): bash tools/prof_build.sh <tag>
tag=${1:-r02}
out=$PWD/gpurun_out/${tag}_build
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c3 -- python3 $OLDPWD/bench.py --config c3 --steps 1 --warmup 0 > $out/c3.json 2> $out/c3.err )
echo "rc=$?"
f=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; head -30 "$f" | cut -c1-260; else echo "no kernel_stats.csv"; ls -R $out | head -20; fi
for f in $(find $out -name "*kernel_trace.csv"); do head -1 $f > $f.tmp; grep -E "k_greedy_search" $f >> $f.tmp; mv $f.tmp $f; done
