# PMC passes over tools/pmc_cmp.py (search kernel vs gather probe); prints per-kernel counter averages
export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmccmp_$i -o p -- python3 tools/pmc_cmp.py > gpurun_out/pmccmp_$i.log 2>&1
  grep -E "probe ms|rror" gpurun_out/pmccmp_$i.log | tail -2
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmccmp_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_greedy_search" in k and "8192" in k: k = "K2"
        elif "gather_probe" in k: k = "probe"
        else: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        for c, v in acc[k].items():
            v = v[-3:]
            print(k, c, sum(v) / len(v))
PY
rm -rf gpurun_out/pmccmp_*/
