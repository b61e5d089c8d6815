#!/bin/bash
# rows phase: persistent waves striding over the patches (default) against one wave per four patches (GPC_SPARSE_ROWS_ALL=1), same library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_${1:-rall}.so
for rep in 1 2 3; do
  for v in off on; do
    if [ $v = on ]; then export GPC_SPARSE_ROWS_ALL=1; else unset GPC_SPARSE_ROWS_ALL; fi
    for w in c4defaults c4defaults3; do
      timeout -k 10 300 python bench.py --only $w --no-cpu-baseline > $O/${w}_ra${v}_${rep}.json 2> $O/${w}_ra${v}_${rep}.err; echo -n "$w all=$v $rep rc=$? "
      python - <<PY
import json
r=json.load(open("$O/${w}_ra${v}_${rep}.json"))
r=r[0] if isinstance(r,list) else r
print(round(r["value"],1), round(r["roofline"]["kernel_ms_stats"]["median"],4), r["config"].get("results_ok"))
PY
    done
  done
done
