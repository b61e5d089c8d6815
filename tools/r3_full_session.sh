#!/bin/bash
# one GPU session: the whole GPU suite, the default bench line, the 1-rank rehearsal of the N > 1 path
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest_full.log | tail -6
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo bench rc=$?
GPC_BENCH_FORCE_DIST=1 python bench.py --steps 5 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo dist rc=$?; tail -3 $O/bench_dist1.err
