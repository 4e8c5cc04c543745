"""Turns the rocprofv3 outputs of one round into the committed summaries under profiles/.

usage: python tools/make_profile_summary.py <tag> <gpurun_out dir> <bench log with the JSON line>
expects in <dir>: <tag>_trace/**/kernel_stats.csv + kernel_trace.csv, <tag>_pmc_fetch/**/counter_collection.csv,
<tag>_pmc_write/**/counter_collection.csv, pmc_expected.json (written by tools/pmc_run.py)."""
import csv
import glob
import json
import statistics as st
import sys

tag, src, benchlog = sys.argv[1], sys.argv[2].rstrip("/") + "/", sys.argv[3]


def one(pattern):
    g = glob.glob(src + pattern, recursive=True)
    assert g, pattern
    return g[0]


bench = None
for line in open(benchlog):
    if line.startswith("{") and '"metric"' in line:
        bench = json.loads(line)
rows = list(csv.DictReader(open(one(tag + "_trace/**/*kernel_stats.csv"))))
with open("profiles/%s_kernel_stats.csv" % tag, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-host-rates (MI355X)\n")
    f.write("# whole process: data generation + K4 build of 1M x 384 + 10 recall batches + 5 warmup + 20 timed steps + 20 counter-pass batches\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows[:14]:
        f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                             r["Percentage"], r["MinNs"], r["MaxNs"]))
tr = list(csv.DictReader(open(one(tag + "_trace/**/*kernel_trace.csv"))))
# the batch kernel only: the build's warm-up rounds run k_greedy_search_wide, 64 queries x 1 024 threads = the same grid size
gs = [r for r in tr if "k_greedy_search<" in r["Kernel_Name"] and r["Grid_Size_X"] == "65536"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in gs]
kname = gs[0]["Kernel_Name"][:120]
exp = json.load(open(src + "pmc_expected.json"))
f1 = list(csv.DictReader(open(one(tag + "_pmc_fetch/**/*counter_collection.csv"))))
f2 = list(csv.DictReader(open(one(tag + "_pmc_write/**/*counter_collection.csv"))))


def vals(rs, kn, name, grid=None):
    return [float(r["Counter_Value"]) for r in rs
            if kn in r["Kernel_Name"] and r["Counter_Name"] == name and (grid is None or r["Grid_Size"] == grid)]


cal_fetch = vals(f1, "k_index_distance", "FETCH_SIZE")[0]
s_fetch = vals(f1, "k_greedy_search<", "FETCH_SIZE", "65536")
s_write = vals(f2, "k_greedy_search<", "WRITE_SIZE", "65536")
s_hit = vals(f2, "k_greedy_search<", "TCC_HIT_sum", "65536")
s_miss = vals(f2, "k_greedy_search<", "TCC_MISS_sum", "65536")
factor = exp["calibration"]["bytes"] / (cal_fetch * 1024)
alg = st.mean(r["alg_bytes"] for r in exp["search"])
read_b = st.mean(s_fetch) * 1024 * factor
write_b = st.mean(s_write) * 1024
# launch order in bench.py: recall batches, warmup, the timed steps, then the counter pass (trace on)
n_recall, n_warm, n_steps = 10, bench["warmup"], bench["steps"]
timed = d[n_recall + n_warm:n_recall + n_warm + n_steps]
t20 = st.mean(timed)
md = f"""# {tag} -- search kernel (K2) profile

Commands (GPU box, `cd /tmp; export TMPDIR=/tmp`):

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/{tag}_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-host-rates
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/{tag}_pmc_fetch -- python3 tools/pmc_run.py
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/{tag}_pmc_write -- python3 tools/pmc_run.py

## `{kname}`
grid 1024 x 64 (one wavefront per query), 1M x 384 cosine, searchSize 75, batch 1024

kernel-trace durations of the {len(d)} batch launches (10 recall + {n_warm} warmup + {n_steps} timed + the counter pass), ms:
{[round(x, 3) for x in d]}

* rocprofv3 average over the {len(timed)} timed launches (trace counters off, batches never walked before): **{t20:.4f} ms**
* bench.py HIP-event average, un-profiled run of the same command: {bench['roofline']['kernel_ms_avg']} ms
  (value {bench['value']} queries/s, roofline.achieved {bench['roofline']['achieved']} GB/s)
* VGPR_Count {gs[0].get('VGPR_Count')}, LDS_Block_Size {gs[0].get('LDS_Block_Size')}, Scratch_Size {gs[0].get('Scratch_Size')}

## HBM traffic (PMC), per launch

Calibration as MI355X_MICROARCH.md (HBM section) prescribes: `k_index_distance` reads {exp['calibration']['rows']} random
slab rows of 1536 B with the same 16 B/lane half-wave loads as the search kernel = {exp['calibration']['bytes']} B known;
FETCH_SIZE reported {cal_fetch:.1f} KB -> correction factor **{factor:.3f}** (the guide's x2 for 16 B/lane streams).

| quantity | value |
|---|---|
| algorithmic bytes / launch (n_dist*1536 + n_edges*4; device counters = oracle counters) | {alg:.4e} |
| FETCH_SIZE / launch (KB, mean of {len(s_fetch)}) | {st.mean(s_fetch):.1f} |
| corrected HBM read bytes / launch (x1024 x{factor:.3f}) | {read_b:.4e} |
| WRITE_SIZE / launch (KB) -> bytes | {st.mean(s_write):.1f} -> {write_b:.4e} |
| read traffic / algorithmic | {read_b / alg:.3f} |
| L2 hit rate TCC_HIT/(HIT+MISS) | {st.mean(s_hit) / (st.mean(s_hit) + st.mean(s_miss)):.4f} |

Achieved (algorithmic bytes / rocprof duration): {alg / (t20 * 1e-3) / 1e9:.0f} GB/s = {alg / (t20 * 1e-3) / 8e12:.3f} of the 8 TB/s peak
({alg / (t20 * 1e-3) / 6.29e12:.3f} of the 6.29 TB/s streaming-copy ceiling of the microarch guide).
"""
open("profiles/%s_search_kernel.md" % tag, "w").write(md)
json.dump({"workload_n": exp["n"], "dim": exp["dim"], "dist": "latent:24", "hbm_bytes_per_launch": int(read_b + write_b),
           "read_bytes": int(read_b), "write_bytes": int(write_b), "fetch_correction": round(factor, 3),
           "source": "profiles/%s_search_kernel.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % tag},
          open("profiles/pmc_traffic.json", "w"), indent=1)
print(md)
