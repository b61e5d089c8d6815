#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for seed in 1 2 3 4; do
  timeout -k 10 600 python tools/r4_stress_sparse.py 500 $seed > $O/stress_sparse_$seed.json 2> $O/stress_sparse_$seed.err; echo "seed $seed rc=$?"; cat $O/stress_sparse_$seed.json | tr -d '\n' | cut -c1-400; echo
done
