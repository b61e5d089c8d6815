#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_var2.so timeout -k 10 600 python -m pytest tests/test_dense_gpu.py -q -m gpu -k "vs_oracle or edge_cases or variance" 2>&1 | tail -2
for lib in base var2 base var2; do
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
  timeout -k 10 300 python bench.py --only c2var 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c2var $lib', round(r['value'],1), 'kernel_ms', round(r['roofline']['kernel_ms'],3), r['roofline']['frac'], r['config']['results_ok'])"
done
