#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of tools/profile_session.sh from gpurun_out/prof_<tag>/ into profiles/ (tracked) and
derives profiles/traffic.json (HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE is in KB and reads exactly 1/2 of a coalesced stream on gfx950 -> x2; WRITE_SIZE is exact)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the most recent run of each pass"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm"):
    for f in newest(os.path.join(src, d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[name] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
summary = {}
for name, cs in pmc.items():
    if "dense_" not in name and "sparse_" not in name:
        continue
    summary[name] = {"launch": meta[name], "counters_mean_per_launch": {k: sum(v) / len(v) for k, v in cs.items()}, "launches": {k: len(v) for k, v in cs.items()}}
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
avg_ns = {}
for r in csv.DictReader(open(stats)):
    avg_ns[r["Name"]] = float(r["AverageNs"])
traffic = {}
for name, s in summary.items():
    c = s["counters_mean_per_launch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        short = "dense_mfma_nt16" if "dense_mfma_kernel<16>" in name else name
        traffic[short] = {"hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                          "fetch_size_kb_raw": c["FETCH_SIZE"], "write_size_kb_raw": c["WRITE_SIZE"],
                          "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
                          "kernel_avg_ns_rocprof": avg_ns.get(name), "source": f"profiles/{tag}_pmc_summary.json"}
with open(os.path.join(dst, "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(traffic, indent=1))
