#!/bin/bash
# second rows phase (two patches per wave, 24 rows of state, patches by ticket) in place of the one-wave small-basis kernel:
# sparse + probit suites on the variant, then alternating C4 timings against the shipped library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
B=${1:-rows5}
V=$PWD/gp_compressor_amd/libgpc_hip_$B.so
GPC_LIB_PATH=$V timeout -k 10 900 python -m pytest tests/test_sparse_gpu.py tests/test_probit_gpu.py -q -m gpu > $O/pytest43_$B.log 2>&1; echo "pytest($B) rc=$?"; tail -5 $O/pytest43_$B.log | cut -c1-200
for rep in 1 2; do
  for v in base $B; do
    if [ $v = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$V; fi
    for w in c4defaults c4defaults3 c4fill; do
      timeout -k 10 300 python bench.py --only $w --no-cpu-baseline > $O/${w}_${v}_${rep}.json 2> $O/${w}_${v}_${rep}.err; echo -n "$w $v $rep rc=$? "
      python - <<PY
import json
r=json.load(open("$O/${w}_${v}_${rep}.json"))
r=r[0] if isinstance(r,list) else r
print(round(r["value"],1), round(r["roofline"]["kernel_ms_stats"]["median"],4), r["config"].get("results_ok"))
PY
    done
  done
done
