#!/bin/bash
# triangular mode of the sparse add (sp_tri_pass): the sparse GPU suite, then same-box A/B of the C4 records against the full passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3tri; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_sparse_gpu.py -m gpu -x -q 2>&1 | tee $O/pytest_sparse.log | tail -15
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
run() { python bench.py --only $1 --steps ${3:-2} 2>$O/err_$1_$2.log > $O/rec_$1_$2.json; python -c "import sys,json; r=json.load(open('$O/rec_$1_$2.json')); print('$1 $2', round(r['ms_per_step'],3), round(r['value']), r['config']['results_ok'], r['roofline']['frac'])"; }
run c4fill tri; GPC_SPARSE_FULL=1 run c4fill full
run c4fill tri; GPC_SPARSE_FULL=1 run c4fill full
run c4defaults tri 5; GPC_SPARSE_FULL=1 run c4defaults full 5
GPC_SPARSE_TRI_MIN=32 run c4defaults tri32 5
GPC_SPARSE_TRI_MIN=32 run c4fill tri32
GPC_SPARSE_TRI_MIN=160 run c4fill tri160
