# usage (GPU box, repo root): bash tools/prof_pq_call.sh <tag>  -- kernel sequence of one quantized search call
# (tools/kernel_ab.py, ROWS_PQ rows x 768, M = 8): what fills the time between the call's kernels
tag=${1:-pqcall}
export TMPDIR=/tmp
ROWS_PQ=${ROWS_PQ:-500000} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/${tag}_trace -- python3 tools/kernel_ab.py > gpurun_out/${tag}.log 2>&1
t=$(find gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_greedy_search<sdb::PQDist" in r["Kernel_Name"]]
last = idx[-1]
seq = rows[idx[-2] + 1:last + 3]
t0 = int(seq[0]["Start_Timestamp"])
for r in seq:
    print("%8.1f us  +%7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
P
find gpurun_out/${tag}_trace -name "*.csv" -size +1M -delete
