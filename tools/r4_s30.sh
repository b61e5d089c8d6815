#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_pfk.so timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py tests/test_host_gpu.py tests/test_producer_gpu.py -q -m gpu 2>&1 | tail -3
for w in c4defaultss c4defaults3 c4fills; do
 for lib in base pfk base pfk; do
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
  GPC_C4_STEPS=2 timeout -k 10 300 python bench.py --only $w 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$w $lib', round(r['value'],1), 'predict_ms', round(r['config']['predict_ms'],3), r['config']['results_ok'])"
 done
done
