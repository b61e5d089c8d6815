#!/usr/bin/env python3
"""Derives profiles/r04_* and profiles/traffic.json from gpurun_out/prof_r04/ (tools/profile_r04.sh).  Merges per workload.

Per workload: rocprofv3's own kernel_stats.csv is copied as profiles/r04_<workload>_kernel_stats.csv; the per-dispatch kernel trace
gives MEDIAN / mean / min / max over the TIMED launches only (the first `warm` dispatches of each kernel are the bench's untimed
warm-up and are dropped) -- for C2 the trace is of `bench.py --steps 20 --warmup 5`, the command the driver times, so the median can be
set beside the bench line's own HIP-event median; the PMC passes give HBM bytes per launch, corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE is in KB and reads exactly 1/2 of a coalesced 16-B-per-lane stream on gfx950 -> x2; WRITE_SIZE is exact),
and, for the sparse workloads, the wave-instructions the add kernels issue per pass (SQ_INSTS_VALU + SALU + LDS + SMEM + VMEM),
which bench.py's issue-rate roofline of the reference-default regime is computed from.
Everything lands in profiles/r04_summary.json; profiles/traffic.json feeds bench.py."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r04")
DST = os.path.join(ROOT, "profiles")

# workload -> (substring of the dominant kernel family, warm-up dispatches per kernel name in the TRACE command, dispatches per kernel
#              name that make up ONE bench launch / pass, traffic.json key, point updates of one pass (sparse) or None)
WORK = {
    "c2": ("dense_w1_kernel", 5, 1, "dense_mfma_w1", None),
    "c2var": ("dense_variance_kernel<16,", 1, 1, "dense_variance@C2", None),
    "c3": ("dense_w1_kernel<512>", 1, 1, "dense_mfma_w1_512", None),
    "c4fill": ("sparse_add_", 4, 4, "sparse_add@C4_fill", 32768 * 256),      # one pass = 4 add calls (rows + small-basis + regular kernel each)
    "c4defaults": ("sparse_add_", 4, 4, "sparse_add@C4_defaults", 32768 * 256),
    "c4defaults3": ("sparse_add_", 4, 4, "sparse_add@C4_defaults_ny3", 32768 * 256),
    "c4fills": ("sparse_predict_", 2, 1, "sparse_predict_sigma@C4_fill", None),        # warm-up (with sigma), mean-only pass, then the sigma pass
    "c4defaultss": ("sparse_predict_", 2, 1, "sparse_predict_sigma@C4_defaults", None),
    "c5": ("dense_big_kernel<8, 1024, 2, 2, true, 4, 3>", 1, 1, "dense_mfma_big_irls@n1024", None),
}
INSTS = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM")


def one(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def main():
    os.makedirs(DST, exist_ok=True)
    summary, traffic = {}, {}
    spath = os.path.join(DST, "r04_summary.json")
    if os.path.exists(spath):
        summary = json.load(open(spath))
    tpath = os.path.join(DST, "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    for w, (kname, warm, per_launch, tkey, pts) in WORK.items():
        base = os.path.join(SRC, w)
        stats = one(os.path.join(base, "trace", "out", "**", "*_kernel_stats.csv"))
        trace = one(os.path.join(base, "trace", "out", "**", "*_kernel_trace.csv"))
        if not stats or not trace:
            continue
        shutil.copy(stats, os.path.join(DST, f"r04_{w}_kernel_stats.csv"))
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        rec = {"kernels": {}, "trace_command": open(os.path.join(SRC, "session.log")).read().split(f"=== {w}/trace: ")[-1].split("\n")[0]
               if os.path.exists(os.path.join(SRC, "session.log")) else None}
        fam_timed = []          # per pass: the summed duration of the family's timed dispatches
        for name, ds in per.items():
            ds.sort()
            dur = [d / 1e6 for _, d in ds]
            k = {"dispatches": len(dur), "avg_ms_all": sum(dur) / len(dur)}
            if kname in name:
                timed = dur[warm:] if len(dur) > warm else dur
                k["warmup_dispatches_dropped"] = min(warm, len(dur))
                k["timed_ms"] = {"median": statistics.median(timed), "mean": sum(timed) / len(timed), "min": min(timed), "max": max(timed),
                                 "dispatches": len(timed)}
                k["avg_ms_timed"] = k["timed_ms"]["mean"]
                passes = max(1, len(timed) // per_launch)
                k["sum_ms_timed_per_bench_launch"] = sum(timed) / passes
                fam_timed.append([sum(timed[i * per_launch:(i + 1) * per_launch]) for i in range(passes)])
            if "gpc" in name or "dense" in name or "sparse" in name or "pc_" in name:
                rec["kernels"][name] = k
        if fam_timed:
            npass = min(len(x) for x in fam_timed)
            tot = [sum(x[i] for x in fam_timed) for i in range(npass)]
            rec["family_ms_per_pass"] = {"median": statistics.median(tot), "mean": sum(tot) / len(tot), "min": min(tot), "max": max(tot),
                                         "passes": npass}
        counters = collections.defaultdict(lambda: collections.defaultdict(list))
        launch_meta = {}
        for p in ("fetch", "write", "sq", "grbm", "sqa", "sqb", "sqc"):
            f = one(os.path.join(base, p, "out", "**", "*_counter_collection.csv"))
            if not f:
                continue
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    counters[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    launch_meta[r["Kernel_Name"]] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                                        "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if k in r}
        if w in ("c4fills", "c4defaultss"):
            # three predict dispatches per command (warm-up, mean-only pass, sigma pass): the counters of the LAST one are the sigma pass's
            for cs in counters.values():
                for c in list(cs):
                    cs[c] = cs[c][-1:]
        rec["launch"] = launch_meta
        rec["counters_mean_per_dispatch"] = {n: {c: sum(v) / len(v) for c, v in cs.items()} for n, cs in counters.items()}
        tr = traffic.get(tkey, {})
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
            tr.update({"hbm_bytes_per_launch": rec["hbm_bytes_per_bench_launch"], "hbm_read_bytes": rec["hbm_read_bytes"],
                       "hbm_write_bytes": rec["hbm_write_bytes"],
                       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B); "
                                     "all dispatches of the kernel family in one bench launch, averaged over the profiled launches "
                                     "(warm-up included: the counters do not depend on clocks)"})
        if "family_ms_per_pass" in rec:
            tr["kernel_ms_timed_rocprof"] = rec["family_ms_per_pass"]["median"]
            tr["kernel_ms_timed_rocprof_stats"] = rec["family_ms_per_pass"]
        # wave-instructions per pass (sparse): every counted instruction class of every add kernel, per pass
        if pts:
            tot_i, tot_v, have = 0.0, 0.0, 0
            for cs in counters.values():
                for c in INSTS:
                    v = cs.get(c, [])
                    if v:
                        tot_i += sum(v) / max(1, len(v) // per_launch)
                        if c == "SQ_INSTS_VALU":
                            tot_v += sum(v) / max(1, len(v) // per_launch)
                        have += 1
            if have and tot_v:
                tr["valu_insts_per_point_update"] = tot_v / pts      # (a 64-lane FP64 operation occupies the SIMD's VALU for four cycles)
            if have:
                rec["wave_insts_per_pass"] = tot_i
                tr["wave_insts_per_launch"] = tot_i
                tr["wave_insts_per_point_update"] = tot_i / pts
                tr["wave_insts_what"] = "sum of " + " + ".join(INSTS) + " over the add kernels of one pass (4 add calls), per point update of the pass"
        tr["source"] = "profiles/r04_summary.json"
        traffic[tkey] = tr
        summary[w] = rec
    json.dump(summary, open(spath, "w"), indent=1)
    json.dump(traffic, open(tpath, "w"), indent=1)
    for w, r in summary.items():
        print(w, json.dumps(r.get("family_ms_per_pass")), "bytes/launch", r.get("hbm_bytes_per_bench_launch"), "insts/pass", r.get("wave_insts_per_pass"))
    print("sections:", sorted(summary))


if __name__ == "__main__":
    main()
