#!/bin/bash
# rocprofv3 evidence for the tiled left-looking kernel (256 < n <= 1024): kernel stats + HBM traffic at the C3 / C5 patch sizes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_big
rm -rf $OUT; mkdir -p $OUT
for n in 512 1024; do
  P=$((2097152 / n / (n / 512)))
  B="python3 bench.py --points $n --patches $P --steps 3 --warmup 1 --no-cpu-baseline"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$n -- $B > $OUT/bench_$n.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$n -- $B > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write$n -- $B > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq$n -- $B > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json
out = {}
for n in (512, 1024):
    P = 2097152 // n // (n // 512)
    rec = {"patches": P, "points": n}
    for f in glob.glob("$OUT/stats%d/*/*_kernel_stats.csv" % n):
        for r in csv.DictReader(open(f)):
            if "big" in r["Name"]:
                rec["kernel"] = r["Name"]; rec["calls"] = int(r["Calls"]); rec["avg_ms"] = float(r["AverageNs"]) / 1e6
    for d in ("fetch", "write", "sq"):
        for f in glob.glob("$OUT/%s%d/*/*_counter_collection.csv" % (d, n)):
            acc = {}
            for r in csv.DictReader(open(f)):
                if "big" in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                rec[k] = sum(v) / len(v)
    if "FETCH_SIZE" in rec:
        rec["hbm_read_GB_per_launch"] = 2 * rec["FETCH_SIZE"] * 1024 / 1e9      # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        rec["hbm_write_GB_per_launch"] = rec.get("WRITE_SIZE", 0) * 1024 / 1e9
        rec["hbm_read_TBps"] = rec["hbm_read_GB_per_launch"] / rec["avg_ms"]
    flops = (3.5 * n * n + n ** 3 / 3 + 2 * n * n + 9 * n * 400) * P
    rec["tflops"] = flops / (rec["avg_ms"] * 1e-3) / 1e12
    rec["frac_of_78.6"] = rec["tflops"] / 78.6
    out["n%d" % n] = rec
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
