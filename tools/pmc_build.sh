# HBM traffic of the build's kernels (rocprofv3 --pmc FETCH_SIZE, its own pass as the guide prescribes), for the
# LDS-tiled prune of new nodes (default) and the one-wave kernel it replaced (BENCH_NO_TILE=1).
# usage (GPU box, repo root): bash tools/pmc_build.sh <tag>
tag=${1:-r02}
export TMPDIR=/tmp
for v in tiled onewave; do
  out=$PWD/gpurun_out/${tag}_pmc_build_$v
  mkdir -p $out
  if [ $v = onewave ]; then export BENCH_NO_TILE=1; else unset BENCH_NO_TILE; fi
  ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof -o c3 -- python3 $OLDPWD/bench.py --config c3 --steps 1 --warmup 0 > $out/c3.json 2> $out/c3.err )
  echo "$v rc=$?"
  python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*counter_collection.csv", recursive=True)
acc, calls = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
    acc[k] += float(r["Counter_Value"])
    calls[k] += 1
top = {k: {"fetch_size_KB_sum": round(v), "x2_GB": round(v * 1024 * 2 / 1e9, 1), "launches": calls[k]} for k, v in acc.most_common(8)}
json.dump(top, open(out + "/fetch_by_kernel.json", "w"), indent=1)
print(json.dumps(top, indent=1))
PY
  rm -rf $out/prof
done
