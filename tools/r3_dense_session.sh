#!/bin/bash
# one GPU session: dense parity tests (+ the flows that use the tiled kernel), C3 / C5 bench records
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_dense_gpu.py tests/test_probit_gpu.py tests/test_producer_gpu.py -m gpu -x -q 2>&1 | tee $O/pytest_dense.log | tail -6
python bench.py --only c3 > $O/c3.json 2> $O/c3.err; echo c3 rc=$?
python bench.py --only c5 > $O/c5.json 2>> $O/c3.err; echo c5 rc=$?
python tools/bench_variance.py > $O/variance.json 2>> $O/c3.err; echo var rc=$?
