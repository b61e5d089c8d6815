#!/bin/bash
# Round 4, GPU session 10: the fixed test, then the dense and sparse suites with every CU's LDS and the workspace poisoned before each kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k small_basis_predict -s > $O/pytest10a.log 2>&1; echo "rc=$?"; grep -E "small-basis predict vs|passed|failed" $O/pytest10a.log | tail -4
GPC_POISON_LDS=1 timeout -k 10 900 python -m pytest tests/test_dense_gpu.py tests/test_sparse_gpu.py tests/test_probit_gpu.py -q -m gpu > $O/pytest10b.log 2>&1; echo "poison rc=$?"; tail -4 $O/pytest10b.log
