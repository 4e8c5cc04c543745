# SQ counters of the exact-scan kernels (rocprofv3 --pmc with --kernel-trace for the durations; no other tracing).
# usage (GPU box, repo root): METRIC=euclidean|cosine bash tools/pmc_scan.sh <tag>   -> gpurun_out/<tag>_pmc_scan_<metric>.json
tag=${1:-r03}
export TMPDIR=/tmp
m=${METRIC:-euclidean}
out=$PWD/gpurun_out/${tag}_pmc_scan_$m
mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  ( cd /tmp && REPS=2 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/prof$i -o scan -- python3 $OLDPWD/tools/bench_flat_scan.py > $out/run$i.log 2>&1 )
  echo "pass $i rc=$?"
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/prof*")):
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
            if k.startswith("k_flat_scan"):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
            if k.startswith("k_flat_scan"):
                per[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in per.items():
            res[k][c] = v[-1]          # the last launch: the longest segment of the scan
    for k, v in dur.items():
        res[k]["last_launch_ms"] = v[-1]
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/prof*
