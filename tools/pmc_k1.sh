# HBM read traffic of the K1 tile kernels (rocprofv3 --pmc FETCH_SIZE, a pass of its own).
# usage (GPU box, repo root): bash tools/pmc_k1.sh <tag>   -> gpurun_out/<tag>_pmc_k1.json
tag=${1:-r03}
export TMPDIR=/tmp
out=$PWD/gpurun_out/${tag}_pmc_k1
mkdir -p $out
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof -o k1 -- python3 $OLDPWD/tools/pmc_k1.py $out/expected.json > $out/run.log 2>&1 )
echo "rc=$?"
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
exp = json.load(open(out + "/expected.json"))
f = glob.glob(out + "/prof/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE":
        per[r["Kernel_Name"].split("(")[0].replace("void sdb::", "")].append(float(r["Counter_Value"]))
cal = [v for k, v in per.items() if k.startswith("k_index_distance")][0][-1]
factor = exp["calibration"]["bytes"] / (cal * 1024)
res = {"shape": "%d x %d x %d" % (exp["nq"], exp["nc"], exp["dim"]), "fetch_correction_factor": round(factor, 3),
       "unique_bytes_(nq+nc)*d*4": exp["unique_bytes"], "pairs_bytes_nq*nc*d*4": exp["pairs_bytes"]}
for k, v in per.items():
    if k.startswith("k_k1_") or k.startswith("k_distance_batch"):
        hbm = sum(v[-3:]) / 3 * 1024 * factor
        res[k] = {"hbm_read_bytes_per_launch": round(hbm), "over_unique_bytes": round(hbm / exp["unique_bytes"], 3),
                  "of_pairs_bytes": round(hbm / exp["pairs_bytes"], 5)}
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/prof
