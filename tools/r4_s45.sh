#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_${1:-mid}.so
timeout -k 10 900 python -m pytest tests/test_sparse_gpu.py tests/test_probit_gpu.py tests/test_host_gpu.py -q -m gpu > $O/pytest45.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest45.log | cut -c1-220
