#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2; do
echo "plain"; timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "none"; LIKE_BENCH=none timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "null + bench"; NULL_STREAM=1 LIKE_BENCH=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "null + bench, one stream"; NULL_STREAM=1 LIKE_BENCH=1 GPC_HOST_ONE_STREAM=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
done
timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_host_gpu.py tests/test_collective_gpu.py tests/test_probit_gpu.py -q -m gpu 2>&1 | tail -2
