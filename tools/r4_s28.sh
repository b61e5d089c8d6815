#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
GPC_POISON_LDS=1 timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_host_gpu.py -q -m gpu > $O/pytest28.log 2>&1; echo "poison rc=$?"; tail -3 $O/pytest28.log
timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_host_gpu.py tests/test_producer_gpu.py -q -m gpu 2>&1 | tail -2
