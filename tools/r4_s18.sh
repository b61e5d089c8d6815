#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_swp.so timeout -k 10 300 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "small_basis_predict or predict_points" 2>&1 | tail -2
for w in c4defaultss c4defaults3; do
 for lib in base swp swp20 base swp swp20; do
  unset GPC_SP_SMALL_WAVES
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_swp.so; fi
  if [ $lib = swp20 ]; then export GPC_SP_SMALL_WAVES=20; fi
  GPC_C4_STEPS=2 timeout -k 10 300 python bench.py --only $w 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$w $lib', round(r['value'],1), 'predict_ms', round(r['config']['predict_ms'],3), r['config']['results_ok'])"
 done
done
