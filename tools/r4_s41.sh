#!/bin/bash
# rows kernel: DPP moves without a zeroed destination, channel count as a template parameter, one register set for the partial sums --
# sparse suite on the variant, then alternating C4-defaults timings: slot-layout build (HEAD) against the variant
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
A=${1:-slotbase}; B=${2:-rows3}
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$B.so timeout -k 10 900 python -m pytest tests/test_sparse_gpu.py tests/test_probit_gpu.py -q -m gpu > $O/pytest41_$B.log 2>&1; echo "pytest($B) rc=$?"; tail -3 $O/pytest41_$B.log
for rep in 1 2 3; do
  for v in $A $B; do
    export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$v.so
    for w in c4defaults c4defaults3; do
      timeout -k 10 300 python bench.py --only $w --no-cpu-baseline > $O/${w}_${v}_${rep}.json 2> $O/${w}_${v}_${rep}.err; echo -n "$w $v $rep rc=$? "
      python - <<PY
import json
r=json.load(open("$O/${w}_${v}_${rep}.json"))
r=r[0] if isinstance(r,list) else r
print(round(r["value"],1), round(r["roofline"]["kernel_ms_stats"]["median"],4), r["config"].get("results_ok"))
PY
    done
  done
done
