# usage (GPU box, repo root): bash tools/prof_flat.sh <tag>  -- per-kernel times of the exact scan (tools/bench_flat.py)
# and the kernel sequence of the last call (limit 75)
tag=${1:-flat}
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 tools/bench_flat.py > gpurun_out/${tag}.log 2>&1
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -12 "$f" | cut -c1-200; cp "$f" gpurun_out/${tag}_kernel_stats.csv; fi
t=$(find gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python3 - "$t" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sdb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "k_flat_emit" in r["Kernel_Name"]]
seq = rows[last[-2] + 1:last[-1] + 1]
t0 = int(seq[0]["Start_Timestamp"])
for r in seq:
    print("%8.1f us  +%7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:50]))
P
fi
find gpurun_out/${tag}_trace -name "*kernel_trace.csv" -delete
grep rows gpurun_out/${tag}.log
