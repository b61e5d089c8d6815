#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for mode in new legacy new legacy; do
  if [ $mode = legacy ]; then export GPC_BENCH_LEGACY_STREAM=1; else unset GPC_BENCH_LEGACY_STREAM; fi
  GPC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-secondary 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('dist1 $mode', round(r['value'],1), round(r['ms_per_step'],4), r['roofline']['kernel_ms'], r['config']['results_ok'])"
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('n1 $mode', round(r['value'],1), round(r['ms_per_step'],4), r['roofline']['kernel_ms'], r['config']['results_ok'])"
done
