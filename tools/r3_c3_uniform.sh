#!/bin/bash
# uniform-batch fast path of the dense dispatch (one size class known on the host): dense + host suites, C3 with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_host_gpu.py -m gpu -x -q 2>&1 | tail -3
show() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],3), round(r['roofline']['frac'],4), r['config']['kernel'])"; }
for r in 1 2; do
  python bench.py --only c3 2>/dev/null | show uniform
  GPC_NO_UNIFORM=1 python bench.py --only c3 2>/dev/null | show split
done
bash tools/profile_r03.sh c3 > $O/prof_c3.log 2>&1; tail -1 $O/prof_c3.log
