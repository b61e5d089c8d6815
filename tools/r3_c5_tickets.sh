#!/bin/bash
# Newton-loop kernel: patches by ticket against the static deal (GPC_BIG_STATIC=1), C5 record, same box; the probit / IRLS suite first
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_probit_gpu.py tests/test_dense_gpu.py -m gpu -x -q 2>&1 | tail -3
show() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],2), round(r['roofline']['frac'],4), r['config']['results_ok'])"; }
for r in 1 2; do
  python bench.py --only c5 2>/dev/null | show tickets
  GPC_BIG_STATIC=1 python bench.py --only c5 2>/dev/null | show static
done
