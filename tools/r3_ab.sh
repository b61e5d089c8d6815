#!/bin/bash
# A/B of two builds on the SAME box: default library vs gp_compressor_amd/libgpc_hip_head.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
W=${1:-c3}
for rep in 1 2 3; do
  for lib in new head; do
    if [ $lib = head ]; then export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_head.so; else unset GPC_LIB_PATH; fi
    python bench.py --only $W 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$W $lib rep $rep', round(r['roofline']['kernel_ms'],3), round(r['roofline']['frac'],4))"
  done
done
