# effective shader clock of a kernel: GRBM_GUI_ACTIVE over the kernel's duration (one rocprofv3 pass: --pmc + --kernel-trace)
# usage (GPU box, repo root): bash tools/pmc_clock.sh <kernel-name-prefix> -- <python script and args>
pref=$1; shift; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_clock
rm -rf $out; mkdir -p $out
( cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/prof -o p -- python3 $OLDPWD/"$@" > $out/run.log 2>&1 )
echo "rc=$?"
python3 - $out $pref <<'PY'
import csv, glob, sys, collections
out, pref = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for f in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
        if k.startswith(pref):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/prof/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void sdb::", "")
        if k.startswith(pref):
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in dur:
    d = dur[k][-1]
    print(k, "duration ms (last launch) %.3f" % d, {c: v[-1] for c, v in cnt[k].items()},
          "GRBM_GUI_ACTIVE / duration = %.3f GHz" % (cnt[k]["GRBM_GUI_ACTIVE"][-1] / d / 1e6) if "GRBM_GUI_ACTIVE" in cnt[k] else "")
PY
