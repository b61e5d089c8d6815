#!/bin/bash
# non-temporal hint on the backward solve's loads: the dense suite on the shipped library (one-wave kernel: hint on), then same-box A/B of
# the tiled kernel with the hint (libgpc_hip_bghint.so, -DBG_HINT=1) on C3 and C5
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_probit_gpu.py -m gpu -x -q 2>&1 | tail -3
show() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],3), round(r['roofline']['frac'],4), r['config']['results_ok'])"; }
python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | show c2
for r in 1 2; do
  for w in c3 c5; do
    unset GPC_LIB_PATH; python bench.py --only $w 2>/dev/null | show "$w base"
    export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_bghint.so; python bench.py --only $w 2>/dev/null | show "$w hint"
  done
done
