#!/bin/bash
# non-temporal hint on the backward solve's loads (shipped: W1_HINT=1, BG_HINT=1): same-box A/B against a build without it.
# Build the other library first:  python -c "from gp_compressor_amd import build; build.build(lib='gp_compressor_amd/libgpc_hip_nohint.so', extra_flags=('-DW1_HINT=0','-DBG_HINT=0'))"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
show() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],3), round(r['roofline']['frac'],4), r['config']['results_ok'])"; }
for r in 1 2; do
  unset GPC_LIB_PATH; python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | show "c2 hint"
  export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_nohint.so; python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | show "c2 nohint"
  for w in c3 c5; do
    unset GPC_LIB_PATH; python bench.py --only $w 2>/dev/null | show "$w hint"
    export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_nohint.so; python bench.py --only $w 2>/dev/null | show "$w nohint"
  done
done
