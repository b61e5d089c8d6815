#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dense_gpu.py -q -m gpu -x > $O/pytest26.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest26.log
for lib in prev base prev base; do
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
  timeout -k 10 300 python bench.py --only c3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c3 $lib', round(r['value'],1), 'kernel_ms', round(r['roofline']['kernel_ms'],3), r['roofline']['frac'], r['config']['kernel'], r['config']['results_ok'])"
done
for lib in prev base; do
  if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
  timeout -k 10 300 python tools/bench_pipeline.py 2>/dev/null | tail -1 | cut -c1-400
done
