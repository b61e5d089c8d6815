#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lib in base pf6 pf8 base pf6 pf8; do
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
  GPC_C4_STEPS=2 timeout -k 10 300 python bench.py --only c4fills 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c4fills $lib', round(r['value'],1), 'predict_ms', round(r['config']['predict_ms'],3), r['config']['results_ok'], r['roofline_predict']['frac'])"
  P=4096 CAP=200 timeout -k 10 200 python tools/bench_sparse.py 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lik $lib', r['likelihood']['s'], r['likelihood']['gflops'])"
done
