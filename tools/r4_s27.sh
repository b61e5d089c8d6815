#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest27.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest27.log
GPC_POISON_LDS=1 timeout -k 10 600 python -m pytest tests/test_dense_gpu.py -q -m gpu > $O/pytest27b.log 2>&1; echo "poison rc=$?"; tail -3 $O/pytest27b.log
bash tools/profile_r04.sh "c3" > $O/prof27.log 2>&1; tail -2 $O/prof27.log
