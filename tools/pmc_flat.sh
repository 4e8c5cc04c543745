# usage (GPU box, repo root): bash tools/pmc_flat.sh <tag>  -- matrix-pipe busy cycles of the exact scan's kernels
# (rocprofv3 --pmc, its own pass; tools/bench_flat.py).  GRBM_GUI_ACTIVE comes summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs (= 32 x the number of
# 16x16x1 instructions): MfmaUtil = busy / (active / 8 x 1024 SIMDs)
tag=${1:-flatpmc}
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/${tag}_pmc -- python3 tools/bench_flat.py > gpurun_out/${tag}.log 2>&1
f=$(find gpurun_out/${tag}_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_flat_scan" not in k:
        continue
    acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k[:40]] += 1
for k, c in acc.items():
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0)
    print(k, dict(c), "MfmaUtil = %.3f" % (busy / (act / 8 * 1024) if act else 0))
P
find gpurun_out/${tag}_pmc -name "*.csv" -size +2M -delete
