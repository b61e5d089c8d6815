#!/bin/bash
# thread-safety of the two-stream host pipeline, second take (threads released together, 120 + 60 calls): without the ordering, with it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
T=tests/test_dense_gpu.py::test_two_stream_pipeline_beside_another_threads_call
GPC_NO_PIPE_ORDER=1 timeout -k 10 300 python -m pytest $T -q -m gpu > $O/pytest37_off.log 2>&1; echo "ordering off rc=$? (1 expected)"; grep -n "^E " $O/pytest37_off.log | head -5 | cut -c1-200
timeout -k 10 300 python -m pytest $T -q -m gpu --durations=1 > $O/pytest37_on.log 2>&1; echo "ordering on rc=$?"; tail -6 $O/pytest37_on.log | cut -c1-200
