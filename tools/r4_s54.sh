#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "random_sweep or rows_phase" --durations=3 > $O/pytest54.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest54.log | cut -c1-200
timeout -k 10 300 python tools/r4_stress_sparse.py 300 1 41 2>&1 | tr -d '\n' | cut -c1-400; echo
