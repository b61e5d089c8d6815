#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python tools/r4_geo_probe.py > $O/geo_probe.json 2> $O/geo_probe.err; echo "rc=$?"; cat $O/geo_probe.json | tr -d '\n ' | sed 's/},{/}\n{/g'; echo
rm -rf $O/kt_f
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f -- python3 bench.py --only c4defaults --no-cpu-baseline > $O/kt_f.log 2>&1; echo "trace rc=$?"
f=$(find $O/kt_f -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1))
PY
find $O/kt_f -type f ! -name "*kernel_stats.csv" -delete
