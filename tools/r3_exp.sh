#!/bin/bash
# Diagnostic builds of the tiled kernel (-DBG_EXP_*: libgpc_hip_<name>.so, see dense_mfma_big.hip) timed on the C2 headline, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
run() { python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],3), r['config']['kernel'], round(r['roofline']['frac'],4), r['config']['results_ok'])"; }
for rep in 1 2; do
  for v in base ${@:-hot nogram noback nopred hotnogram}; do
    if [ $v = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$v.so; fi
    run $v
  done
done
