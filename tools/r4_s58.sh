#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for seed in 1 5; do
  timeout -k 10 600 python tools/r4_stress_sparse.py 400 $seed > $O/stress_sparse_p$seed.json 2> $O/stress_sparse_p$seed.err; echo "seed $seed rc=$?"; cat $O/stress_sparse_p$seed.json | tr -d '\n' | cut -c1-500; echo
done
timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "random_sweep" > $O/pytest58.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest58.log
