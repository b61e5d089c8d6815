#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for v in noshdiv noreg2; do
  for c in 41 147; do
    GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$v.so timeout -k 10 300 python tools/r4_stress_sparse.py 300 1 $c > $O/stress_${v}_$c.json 2> $O/stress_${v}_$c.err
    python - <<PY
import json
r=json.load(open("$O/stress_${v}_$c.json"))
print("$v", $c, {k:("OK" if not d else "DIFF") for k,d in r["report"].items()})
PY
  done
done
