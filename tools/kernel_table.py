#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in the built libsemadb_amd.so, read from the gfx950 code
objects' own metadata notes (.vgpr_count, .agpr_count, .sgpr_count, .vgpr_spill_count, .private_segment_fixed_size,
.group_segment_fixed_size, .max_flat_workgroup_size) -- the figures DESIGN.md quotes come from here, not from memory.

    python3 tools/kernel_table.py                       # markdown table of the walk / build kernels
    python3 tools/kernel_table.py --all                 # every kernel
    python3 tools/kernel_table.py --json out.json       # machine-readable
    python3 tools/kernel_table.py --check               # exit 1 if ANY kernel has scratch or spilled VGPRs, or a kernel
                                                        # spills more SGPRs than its class allows (tests use this)

Needs only the ROCm LLVM tools (llvm-objdump --offloading, llvm-readelf; binutils c++filt); no GPU."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "semadb_amd", "libsemadb_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# Round 6: NO kernel of the library may carry scratch memory or spilled VGPRs (--check fails on any).
# Spilled SGPRs live in lanes of a VGPR, not in memory; they cost lane moves.  The kernels a BASELINE configuration
# launches inside a timed region (C2 walk and its small-call forms, the C4 quantized walks and their table / encode /
# k-means kernels, the C3 build's prunes and back-edges, the exact scan, the merge) may spill at most SGPR_TIMED of
# them; the filtered forms of the walk carry six more list pointers and get SGPR_FILTERED; everything else SGPR_ANY
# (the run-time-length fallbacks of the quantizer kernels sit at 112 .. 135).
SGPR_TIMED, SGPR_FILTERED, SGPR_ANY = 64, 128, 160
TIMED = re.compile(
    r"^sdb::(k_greedy_search<sdb::PlainDist<(3|6), false, true, 0>, 2, (false|true), 8192u>"
    r"|k_greedy_search_wide<(3|6), false, (8|16), (false|true)>"
    r"|k_greedy_search_pq2<|k_greedy_search_pqw<15, 33, 4294967295u, 4, 15, true>"
    r"|k_pq_lut_t<true, 3, 96>|k_pq_lut_t<true, 0, 4>|k_pq_lut_mfma<|k_pq_encode_t<true, 3, true>|k_pq_encode_pair<true, 4>"
    r"|k_km_assign_t<3, 96>|k_km_assign_t<0, 4>|k_km_(?!assign_t<)|k_backedges<(3|6), false>|k_prune_|k_flat_|k_topk_merge|k_k1_|k_index_distance"
    r"|k_adjcodes|k_filter_)")
FILTERED = re.compile(r"^sdb::(k_greedy_search<.*, true, \d+u>|k_greedy_search_wide<.*, true>)$")

FIELDS = [".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size"]


def code_objects(so, tmp):
    dst = os.path.join(tmp, os.path.basename(so))
    shutil.copy(so, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def kernels(so=SO):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            cur = None
            for line in notes.splitlines():
                if line.startswith("  - ") and cur is not None and ".name" in cur:
                    out.append(cur)
                    cur = {}
                elif line.startswith("  - "):
                    cur = {}
                if cur is None:
                    continue
                m = re.match(r"^(?:  - |    )(\.[a-z_]+):\s+(\S.*)$", line)
                if m and m.group(1) in FIELDS + [".name"]:
                    cur[m.group(1)] = m.group(2).strip()
                if line.startswith("amdhsa.target") or line.startswith("amdhsa.version"):
                    if cur and ".name" in cur:
                        out.append(cur)
                    cur = None
            if cur and ".name" in cur:
                out.append(cur)
    names = "\n".join(k[".name"] for k in out)
    dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True, check=True).stdout.splitlines()
    rows = []
    for k, d in zip(out, dem):
        d = re.sub(r"\s*\[clone .*\]$", "", d)
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*\)$", "", d)  # drop the parameter list
        row = {"kernel": d}
        for f in FIELDS:
            row[f[1:]] = int(k.get(f, "0"))
        rows.append(row)
    rows.sort(key=lambda r: r["kernel"])
    return rows


def waves_per_simd(r):
    # gfx950: 512 VGPRs per SIMD lane shared by arch + acc registers, allocation granule 8, at most 8 waves per SIMD.
    # .vgpr_count of a gfx90a+ code object is the unified total (arch registers up to the accumulation offset + AGPRs)
    regs = r["vgpr_count"]
    regs = max(8, (regs + 7) // 8 * 8)
    return min(8, 512 // regs)


def is_hot(name):
    return bool(TIMED.match(name))


def sgpr_limit(name):
    if FILTERED.match(name):
        return SGPR_FILTERED
    return SGPR_TIMED if is_hot(name) else SGPR_ANY


def offenders(rows):
    return [r for r in rows if r["vgpr_spill_count"] or r["private_segment_fixed_size"] or
            r["sgpr_spill_count"] > sgpr_limit(r["kernel"])]


def markdown(rows):
    lines = ["| kernel | VGPR (of which AGPR) | SGPR | spilled VGPR | spilled SGPR | scratch B | static LDS B | max WG | waves/SIMD by registers |",
             "|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %d (%d) | %d | %d | %d | %d | %d | %d | %d |" % (
            r["kernel"], r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"],
            r["private_segment_fixed_size"], r["group_segment_fixed_size"], r["max_flat_workgroup_size"], waves_per_simd(r)))
    return "\n".join(lines)


# the kernels DESIGN.md's register table shows: what the BASELINE configurations launch in their timed regions
DESIGN = re.compile(
    r"^sdb::(k_greedy_search<sdb::PlainDist<(3|6), false, true, 0>, 2, (false|true), 8192u>"
    r"|k_greedy_search_wide<3, false, (8|16), false>|k_greedy_search_pq2<4294967295u>"
    r"|k_greedy_search_pqw<15, 33, 4294967295u, 4, 15, true>|k_pq_lut_t<true, (3, 96|0, 4)>|k_pq_encode_t<true, 3, true>"
    r"|k_pq_encode_pair<true, 4>|k_km_assign_t<(3, 96|0, 4)>|k_backedges<3, false>|k_prune_new<3, false>"
    r"|k_prune_new_tiled<3, false, false, 8>|k_flat_scan_mfma<12, false>|k_flat_scan<false>|k_k1_stream_mfma<12>"
    r"|k_k1_tile_mfma<12, false>|k_topk_merge)")
BEGIN, END = "<!-- kernel_table:begin (python3 tools/kernel_table.py --design) -->", "<!-- kernel_table:end -->"


def design_block(rows):
    sel = [r for r in rows if DESIGN.match(r["kernel"])]
    worst_t = max([r["sgpr_spill_count"] for r in rows if is_hot(r["kernel"]) and not FILTERED.match(r["kernel"])] or [0])
    worst = max([r["sgpr_spill_count"] for r in rows] or [0])
    head = ("%d kernels in the library; %d with spilled VGPRs or scratch memory; most spilled SGPRs: %d on a timed path, "
            "%d anywhere.\n\n" % (len(rows), sum(1 for r in rows if r["vgpr_spill_count"] or r["private_segment_fixed_size"]),
                                  worst_t, worst))
    return BEGIN + "\n" + head + markdown(sel) + "\n" + END


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=SO)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--match", default=None, help="regular expression on the demangled name")
    ap.add_argument("--json", default=None)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--design", action="store_true", help="the block DESIGN.md carries between its kernel_table markers")
    a = ap.parse_args()
    rows = kernels(a.so)
    if a.design:
        print(design_block(rows))
        return 0
    if a.check:
        bad = offenders(rows)
        for r in bad:
            print("SPILL %s: %d VGPRs spilled, %d B scratch, %d SGPRs spilled (limit %d)" % (
                r["kernel"], r["vgpr_spill_count"], r["private_segment_fixed_size"], r["sgpr_spill_count"], sgpr_limit(r["kernel"])))
        print("%d kernels, %d on timed paths, %d with scratch, spilled VGPRs or too many spilled SGPRs; most SGPRs spilled: "
              "%d on a timed path, %d anywhere" % (
                  len(rows), sum(is_hot(r["kernel"]) for r in rows), len(bad),
                  max([r["sgpr_spill_count"] for r in rows if is_hot(r["kernel"]) and not FILTERED.match(r["kernel"])] or [0]),
                  max([r["sgpr_spill_count"] for r in rows] or [0])))
        return 1 if bad else 0
    sel = rows
    if a.match:
        sel = [r for r in rows if re.search(a.match, r["kernel"])]
    elif not a.all:
        sel = [r for r in rows if re.match(r"^sdb::(k_greedy_search|k_backedges|k_prune_new|k_pq_lut|k_flat_scan|k_k1_)", r["kernel"])]
    if a.json:
        with open(a.json, "w") as f:
            json.dump(sel, f, indent=1)
    print(markdown(sel))
    return 0


if __name__ == "__main__":
    sys.exit(main())
