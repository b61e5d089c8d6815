#!/bin/bash
# kernel trace of the C4-defaults pass with a variant library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
B=${1:-rows5}
export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$B.so
for w in c4defaults c4defaults3; do
  rm -rf $O/kt_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 bench.py --only $w --no-cpu-baseline > $O/kt_$w.log 2>&1; echo "$w rc=$?"
  f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1))
PY
  find $O/kt_$w -type f ! -name "*kernel_stats.csv" -delete
done
