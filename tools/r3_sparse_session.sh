#!/bin/bash
# one GPU session: sparse parity tests, C4 defaults bench (+ variants), per-kernel profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py tests/test_probit_gpu.py tests/test_host_gpu.py -m gpu -x -q 2>&1 | tee $O/pytest_sparse.log | tail -6
python bench.py --only c4defaults > $O/c4d.json 2> $O/c4d.err; echo c4d rc=$?
GPC_SPARSE_NO_LIST=1 python bench.py --only c4fill > $O/c4f_nolist.json 2>> $O/c4d.err
python bench.py --only c4defaults3 > $O/c4d3.json 2>> $O/c4d.err
python bench.py --only c4fill > $O/c4f.json 2>> $O/c4d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4d -- python3 bench.py --only c4defaults > $O/prof_c4d.log 2>&1
find $O/prof_c4d -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150
find $O/prof_c4d -type f ! -name "*stats.csv" -delete
