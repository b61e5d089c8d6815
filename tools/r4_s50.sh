#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python tools/r4_stress_sparse.py ${1:-300} ${2:-1} > $O/stress_sparse.json 2> $O/stress_sparse.err; echo "rc=$?"; tail -4 $O/stress_sparse.err; cat $O/stress_sparse.json | tr -d '\n' | cut -c1-600; echo
