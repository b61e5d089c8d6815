#!/bin/bash
# C2 headline on the three kernel shapes, same box: register-resident | tiled, 4 waves x 2 workgroups | tiled, 2 waves x 4 workgroups
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
run() { python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['value']), round(r['ms_per_step'],3), r['config']['kernel'], round(r['roofline']['frac'],4), r['config']['results_ok'])"; }
for rep in 1 2; do
  run reg
  GPC_FORCE_BIG=1 GPC_BIG_NO_W2=1 run w4
  GPC_FORCE_BIG=1 run w2
done
