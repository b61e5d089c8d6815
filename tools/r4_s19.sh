#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lib in bm28 bm32; do
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so timeout -k 10 300 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "bit_identical or known_answers or batch_vs_oracle" 2>&1 | tail -2
done
for w in c4defaults; do
 for lib in base bm28 bm32 base bm28 bm32; do
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
  GPC_C4_STEPS=3 timeout -k 10 300 python bench.py --only $w 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$w $lib', round(r['value'],1), 'add_ms', round(r['roofline']['kernel_ms'],3), r['config']['results_ok'], r['config']['bv_max'])"
 done
done
