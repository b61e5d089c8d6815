#!/bin/bash
# the sparse and probit suites with every kernel family's LDS poisoned with NaN beforehand (GPC_POISON_LDS=1): the slot layout of the rows
# kernels relies on explicitly zeroed blocks, a read of a word nobody wrote would be a NaN in a state
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
GPC_POISON_LDS=1 timeout -k 10 900 python -m pytest tests/test_sparse_gpu.py tests/test_probit_gpu.py -q -m gpu > $O/pytest59.log 2>&1; echo "pytest(poison) rc=$?"; tail -4 $O/pytest59.log | cut -c1-200
