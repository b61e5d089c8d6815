#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for rep in 1 2; do
  timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
  GPC_HOST_ONE_STREAM=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
  GPC_HOST_NO_PIPELINE=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
done
P=16384 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
P=16384 GPC_HOST_ONE_STREAM=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
P=8192 N=128 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
P=8192 N=128 GPC_HOST_ONE_STREAM=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1
timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_host_gpu.py tests/test_capi_gpu.py -q -m gpu 2>&1 | tail -3
