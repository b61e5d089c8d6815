#!/bin/bash
# Round 4, GPU session 2: the round-4 build of the one-wave kernel (libgpc_hip_v2.so and its switches) -- dense parity suite on it, then
# same-box timing against the round-3 kernel and the instruction counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
step() { local secs=$1 log=$2; shift 2; echo "=== $* (limit ${secs}s) $(date +%T)" | tee -a $O/session2.log
         timeout -k 10 "$secs" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc $(date +%T)" | tee -a $O/session2.log
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $O/session2.log; exit 1; fi; return $rc; }
V=${1:-v2}
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$V.so step 600 $O/pytest2_$V.log python -m pytest tests/test_dense_gpu.py tests/test_host_gpu.py tests/test_producer_gpu.py tests/test_collective_gpu.py -q -m gpu
tail -15 $O/pytest2_$V.log
shift
bash tools/r3_exp.sh $V "$@" > $O/exp2_time.log 2>&1; cat $O/exp2_time.log
bash tools/r3_exp_pmc.sh base $V > $O/exp2_pmc.log 2>&1; tail -3 $O/exp2_pmc.log
