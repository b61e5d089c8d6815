#!/bin/bash
# end-of-round session: the whole GPU suite, smoke(), the default bench line, the 1-rank rehearsal of the N > 1 path, then the rocprofv3
# passes of the dense workloads (tools/profile_r03.sh; tools/collect_profiles_r03.py derives profiles/ from them)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest_full.log | tail -4
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo bench rc=$?
GPC_BENCH_FORCE_DIST=1 python bench.py --steps 5 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo dist rc=$?
bash tools/profile_r03.sh "${1:-c2 c2var c3 c5}" > $O/prof.log 2>&1; tail -2 $O/prof.log
