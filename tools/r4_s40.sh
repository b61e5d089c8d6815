#!/bin/bash
# rows kernel in the SLOT layout (sp_slot): bit-identity suite on the variant, then alternating C4-defaults timings against the
# shipped library (which is the build BEFORE the select reduction of tools/r4_s39.sh: compare with that session's variant numbers too)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
V=$PWD/gp_compressor_amd/libgpc_hip_slot.so
GPC_LIB_PATH=$V timeout -k 10 900 python -m pytest tests/test_sparse_gpu.py -q -m gpu > $O/pytest40.log 2>&1; echo "pytest(variant) rc=$?"; tail -3 $O/pytest40.log
for rep in 1 2 3; do
  for v in base slot; do
    if [ $v = slot ]; then export GPC_LIB_PATH=$V; else unset GPC_LIB_PATH; fi
    for w in c4defaults c4defaults3; do
      timeout -k 10 300 python bench.py --only $w --no-cpu-baseline > $O/${w}_${v}_${rep}.json 2> $O/${w}_${v}_${rep}.err; echo "$w $v $rep rc=$?"
      python - <<PY
import json
r=json.load(open("$O/${w}_${v}_${rep}.json"))
r=r[0] if isinstance(r,list) else r
print("$w $v", round(r["value"],1), r["roofline"].get("kernel_ms_stats", r["roofline"].get("add_ms_stats")), r["config"].get("results_ok"))
PY
    done
  done
done
