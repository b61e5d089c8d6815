#!/bin/bash
# triangular passes of the sparse add (sp_tri_pass): same-box A/B of a C4 record -- the shipped library (tri), the full passes
# (GPC_SPARSE_FULL=1), optionally another build of the library (gp_compressor_amd/libgpc_hip_prev.so: prev); three rounds, alternating.
# Boxes differ by 10-15 % on this path (the full passes are HBM-bound, the triangular ones are not): only same-box numbers compare.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3tri; mkdir -p $O
W=${W:-c4fill}
run() { python bench.py --only $W --steps 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$W $1', round(r['ms_per_step'],1), round(r['value']), r['config']['results_ok'], round(r['roofline']['frac'],3))"; }
for rep in 1 2 3; do
  for v in ${@:-tri full}; do
    unset GPC_SPARSE_FULL GPC_LIB_PATH
    if [ $v = full ]; then export GPC_SPARSE_FULL=1
    elif [ $v = prev ]; then export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_prev.so; fi
    run $v
  done
done 2>&1 | tee $O/ab_$W.log
