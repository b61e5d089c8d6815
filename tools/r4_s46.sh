#!/bin/bash
# second rows phase with static first tickets (GPC_SPARSE_ROWS2=1) against the default (one-wave kernel), same library; then its kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_${1:-rows2b}.so
timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "rows_phase or small_basis_phase or defaults" > $O/pytest46.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest46.log | cut -c1-200
for rep in 1 2 3; do
  for v in off on; do
    if [ $v = on ]; then unset GPC_SPARSE_NO_ROWS2; else export GPC_SPARSE_NO_ROWS2=1; fi
    for w in c4defaults c4defaults3; do
      timeout -k 10 300 python bench.py --only $w --no-cpu-baseline > $O/${w}_r2${v}_${rep}.json 2> $O/${w}_r2${v}_${rep}.err; echo -n "$w rows2=$v $rep rc=$? "
      python - <<PY
import json
r=json.load(open("$O/${w}_r2${v}_${rep}.json"))
r=r[0] if isinstance(r,list) else r
print(round(r["value"],1), round(r["roofline"]["kernel_ms_stats"]["median"],4), r["config"].get("results_ok"))
PY
    done
  done
done
unset GPC_SPARSE_NO_ROWS2
rm -rf $O/kt_r2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_r2 -- python3 bench.py --only c4defaults --no-cpu-baseline > $O/kt_r2.log 2>&1; echo "trace rc=$?"
f=$(find $O/kt_r2 -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:7]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["MinNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1))
PY
find $O/kt_r2 -type f ! -name "*kernel_stats.csv" -delete
