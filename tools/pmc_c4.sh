# HBM traffic of the quantized search kernels (rocprofv3 --pmc FETCH_SIZE, a pass of its own as the guide prescribes).
# usage (GPU box, repo root): bash tools/pmc_c4.sh <tag>     -> gpurun_out/<tag>_pmc_c4_M<M>.json
tag=${1:-r03}
export TMPDIR=/tmp
for M in 192 8; do
  out=$PWD/gpurun_out/${tag}_pmc_c4_M$M
  mkdir -p $out
  export PMC_M=$M
  ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof -o c4 -- python3 $OLDPWD/tools/pmc_c4.py $out/expected.json > $out/run.log 2>&1 )
  echo "M=$M rc=$?"
  python3 - $out $M <<'PY'
import csv, glob, sys, collections, json
out, M = sys.argv[1], int(sys.argv[2])
exp = json.load(open(out + "/expected.json"))
f = glob.glob(out + "/prof/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
    per[k].append(float(r["Counter_Value"]))
cal = [v for k, v in per.items() if k.startswith("k_index_distance")][0][-1]
factor = exp["calibration"]["bytes"] / (cal * 1024)   # FETCH_SIZE is in KB; the gfx950 correction comes out of the calibration
res = {"M": M, "rows": exp["n"], "dim": exp["dim"], "fetch_correction_factor": round(factor, 3)}
for k, v in per.items():
    if k.startswith("k_greedy_search") and ("PQDist" in k or "pqw" in k or "pq2" in k) and len(v) >= 5:  # not the build's walks
        last = v[-5:]
        rows = []
        for fs, e in zip(last, exp["search"]):
            hbm = fs * 1024 * factor
            rows.append({"hbm_read_bytes": round(hbm), "code_and_edge_bytes": e["code_and_edge_bytes"], "table_bytes": e["table_bytes"],
                         "traffic_over_codes_and_edges": round(hbm / e["code_and_edge_bytes"], 3),
                         "traffic_over_codes_edges_tables": round(hbm / (e["code_and_edge_bytes"] + e["table_bytes"]), 3),
                         "traffic_over_64B_sector_floor": round(hbm / (e["code_rows_at_64B_sectors"] + e["table_bytes"]), 3)})
        res[k] = rows
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
  rm -rf $out/prof
done
