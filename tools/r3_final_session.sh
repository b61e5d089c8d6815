#!/bin/bash
# end-of-round session: the whole GPU suite, the default bench line, the 1-rank rehearsal of the N > 1 path, where the triangular
# passes should start (GPC_SPARSE_TRI_MIN), the C3 trace (classify kernel with one atomic per wave)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest_full.log | tail -4
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo bench rc=$?
GPC_BENCH_FORCE_DIST=1 python bench.py --steps 5 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo dist rc=$?
run() { python bench.py --only $1 --steps ${3:-2} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 $2', round(r['ms_per_step'],3), round(r['value']), r['config']['results_ok'])"; }
for rep in 1 2; do
  for m in 32 0 24; do GPC_SPARSE_TRI_MIN=$m run c4fill trimin=$m; done
  for m in 32 0 24; do GPC_SPARSE_TRI_MIN=$m run c4defaults trimin=$m 5; done
done 2>&1 | tee $O/trimin.log
bash tools/profile_r03.sh "c3" > $O/prof_c3.log 2>&1; tail -2 $O/prof_c3.log
