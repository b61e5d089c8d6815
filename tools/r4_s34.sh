#!/bin/bash
# C5 (IRLS, tiled kernel) with the link functor called out of line (scratch 476 -> 144 B/lane): parity of the probit suite on the variant, then
# alternating timings of the shipped library and the variant on the same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
V=$PWD/gp_compressor_amd/libgpc_hip_ni.so
GPC_LIB_PATH=$V timeout -k 10 600 python -m pytest tests/test_probit_gpu.py -q -m gpu > $O/pytest34.log 2>&1; echo "pytest(variant) rc=$?"; tail -3 $O/pytest34.log
for rep in 1 2 3; do
  for v in base ni; do
    if [ $v = ni ]; then export GPC_LIB_PATH=$V; else unset GPC_LIB_PATH; fi
    timeout -k 10 300 python bench.py --only c5 > $O/c5_${v}_${rep}.json 2> $O/c5_${v}_${rep}.err; echo "c5 $v $rep rc=$?"
    python - <<PY
import json
r=json.load(open("$O/c5_${v}_${rep}.json"))
print("$v", round(r["value"],1), r["roofline"]["frac"], r["config"].get("results_ok"))
PY
  done
done
