#!/usr/bin/env python3
"""Derives profiles/r03_* and profiles/traffic.json from gpurun_out/prof_r03/ (tools/profile_r03.sh).  MERGES: a run that profiled
only some workloads updates only their sections of profiles/r03_summary.json (round 2's collector overwrote the file and lost three).

Per workload: rocprofv3's own kernel_stats.csv is copied as profiles/r03_<workload>_kernel_stats.csv; the per-dispatch
kernel trace gives the average duration over the TIMED launches only (the first `warm` dispatches of each kernel are
the bench's untimed warm-up and are dropped); the PMC passes give HBM bytes per launch, corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE is in KB and reads exactly 1/2 of a coalesced 16-B-per-lane stream on gfx950 -> x2; WRITE_SIZE is
exact).  Everything lands in profiles/r03_summary.json; profiles/traffic.json feeds bench.py's roofline.traffic."""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r03")
DST = os.path.join(ROOT, "profiles")

# workload -> (substring of the dominant kernel's name, warm-up dispatches per kernel name in the profiled command, dispatches
#              per kernel name that make up ONE bench launch / pass, traffic.json key)
WORK = {
    "c2": ("dense_w1_kernel", 2, 1, "dense_mfma_w1"),
    "c2var": ("dense_variance_kernel<16,", 1, 1, "dense_variance@C2"),
    "c3": ("dense_big_kernel<4, 512, 2, 2, false, 4, 1>", 1, 1, "dense_mfma_big@n512"),
    "c4fill": ("sparse_add_", 4, 4, "sparse_add@C4_fill"),          # one pass = 4 add calls, each a small-basis + a regular kernel
    "c4defaults": ("sparse_add_", 4, 4, "sparse_add@C4_defaults"),        # + the rows phase (sparse_add_rows_kernel), same family
    "c4defaults3": ("sparse_add_", 4, 4, "sparse_add@C4_defaults_ny3"),
    "c5": ("dense_big_kernel<8, 1024, 2, 2, true, 4, 3>", 1, 1, "dense_mfma_big_irls@n1024"),
}


def one(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def main():
    os.makedirs(DST, exist_ok=True)
    summary, traffic = {}, {}
    spath = os.path.join(DST, "r03_summary.json")
    if os.path.exists(spath):
        summary = json.load(open(spath))          # merge: sections of workloads not profiled in this run stay
    tpath = os.path.join(DST, "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    for w, (kname, warm, per_launch, tkey) in WORK.items():
        base = os.path.join(SRC, w)
        stats = one(os.path.join(base, "trace", "out", "**", "*_kernel_stats.csv"))
        trace = one(os.path.join(base, "trace", "out", "**", "*_kernel_trace.csv"))
        if not stats or not trace:
            continue
        shutil.copy(stats, os.path.join(DST, f"r03_{w}_kernel_stats.csv"))
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        rec = {"kernels": {}}
        for name, ds in per.items():
            ds.sort()
            dur = [d for _, d in ds]
            k = {"dispatches": len(dur), "avg_ms_all": sum(dur) / len(dur) / 1e6}
            if kname in name:
                timed = dur[warm:] if len(dur) > warm else dur
                k["warmup_dispatches_dropped"] = min(warm, len(dur))
                k["avg_ms_timed"] = sum(timed) / len(timed) / 1e6
                k["sum_ms_timed_per_bench_launch"] = sum(timed) / 1e6 / max(1, len(timed) // per_launch)
            if "gpc" in name or "dense" in name or "sparse" in name or "pc_" in name:
                rec["kernels"][name] = k
        counters = collections.defaultdict(lambda: collections.defaultdict(list))
        launch_meta = {}
        for p in ("fetch", "write", "sq", "grbm"):
            f = one(os.path.join(base, p, "out", "**", "*_counter_collection.csv"))
            if not f:
                continue
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    counters[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    launch_meta[r["Kernel_Name"]] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                                        "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if k in r}
        rec["launch"] = launch_meta
        rec["counters_mean_per_dispatch"] = {n: {c: sum(v) / len(v) for c, v in cs.items()} for n, cs in counters.items()}
        # HBM bytes per bench launch: every dispatch of the dominant kernel family in one pass, summed
        fetch = sum(sum(cs.get("FETCH_SIZE", [])) for cs in counters.values())
        write = sum(sum(cs.get("WRITE_SIZE", [])) for cs in counters.values())
        nf = max([len(cs.get("FETCH_SIZE", [])) for cs in counters.values()] or [0])      # dispatches per kernel name
        nw = max([len(cs.get("WRITE_SIZE", [])) for cs in counters.values()] or [0])
        if nf and nw:
            launches = max(1, nf // per_launch)
            rec["hbm_bytes_per_bench_launch"] = (2.0 * fetch / launches + write / max(1, nw // per_launch)) * 1024.0
            rec["hbm_read_bytes"] = 2.0 * fetch / launches * 1024.0
            rec["hbm_write_bytes"] = write / max(1, nw // per_launch) * 1024.0
            ms = [k.get("sum_ms_timed_per_bench_launch") for n, k in rec["kernels"].items() if kname in n and "sum_ms_timed_per_bench_launch" in k]
            traffic[tkey] = {"hbm_bytes_per_launch": rec["hbm_bytes_per_bench_launch"], "hbm_read_bytes": rec["hbm_read_bytes"],
                             "hbm_write_bytes": rec["hbm_write_bytes"],
                             "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B); "
                                           "all dispatches of the kernel family in one bench launch, averaged over the profiled launches "
                                           "(warm-up included: the counters do not depend on clocks)",
                             "kernel_ms_timed_rocprof": sum(ms) if ms else None, "source": "profiles/r03_summary.json"}
        summary[w] = rec
    json.dump(summary, open(spath, "w"), indent=1)
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps({w: {n: {kk: vv for kk, vv in k.items()} for n, k in r["kernels"].items()} for w, r in summary.items()}, indent=1))
    print("sections:", sorted(summary))
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
