#!/bin/bash
# Round 4, GPU session 6: the small-basis predict kernel (libgpc_hip_<variant>.so): sparse / host / producer suites on it, then the C4 records
# with and without sigma against the shipped library on the same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
V=${1:-sp2}
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$V.so timeout -k 10 900 python -m pytest tests/test_sparse_gpu.py tests/test_host_gpu.py tests/test_producer_gpu.py tests/test_probit_gpu.py -q -m gpu > $O/pytest6.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest6.log
for w in c4defaultss c4fills c4defaults3; do
  for lib in base $V base $V; do
    if [ $lib = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$lib.so; fi
    GPC_C4_STEPS=2 timeout -k 10 300 python bench.py --only $w 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$w $lib', round(r['value'],1), 'ms/step', round(r['ms_per_step'],3), 'predict_ms', round(r['config']['predict_ms'],3), r['config']['results_ok'], r.get('roofline_predict',{}).get('frac'))"
  done
done
