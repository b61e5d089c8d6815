#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python tools/r4_stress_sparse.py 300 1 ${1:-41} > $O/stress_one.json 2> $O/stress_one.err; echo "rc=$?"; tail -3 $O/stress_one.err; cat $O/stress_one.json | cut -c1-300 | head -80
