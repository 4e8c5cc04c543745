# SQ counters summed over all launches of the kernels whose name starts with <prefix> (rocprofv3 --pmc + --kernel-trace).
# usage (GPU box, repo root): bash tools/pmc_sq.sh <tag> <prefix[,prefix..]> -- <python script and args>
tag=$1; pref=$2; shift; shift; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/${tag}_pmc_sq
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/prof$i -o p -- python3 $OLDPWD/"$@" > $out/run$i.log 2>&1 )
  echo "pass $i rc=$?"
done
python3 - $out $pref <<'PY'
import csv, glob, sys, collections, json
out, prefs = sys.argv[1], sys.argv[2].split(",")
res = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob(out + "/prof*")):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        seen = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
            if any(k.startswith(p) for p in prefs):
                seen[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[k] += 1
        for k in seen:
            res[k]["total_ms"] = seen[k]; res[k]["launches"] = n[k]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
            if any(k.startswith(p) for p in prefs):
                res[k][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/prof*
