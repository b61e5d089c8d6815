#!/bin/bash
# Round 4, GPU session 3: the LDS-resident triangular mode of the sparse add kernel (libgpc_hip_res.so): sparse suite on it, then
# capacity 100 / 120 / 80 in the basis-filling regime against the HBM-resident two-wave shape on the same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
step() { local secs=$1 log=$2; shift 2; echo "=== $* (limit ${secs}s) $(date +%T)" | tee -a $O/session3.log
         timeout -k 10 "$secs" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc $(date +%T)" | tee -a $O/session3.log
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $O/session3.log; exit 1; fi; return $rc; }
export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_${1:-res}.so
step 300 $O/pytest3a.log python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "lds_resident or triangular_mode" -x || { tail -40 $O/pytest3a.log; exit 1; }
tail -3 $O/pytest3a.log
for cap in 100 120 80; do
  for rep in 1 2; do
    P=8192 CAP=$cap step 200 $O/bs_res_$cap.log python tools/bench_sparse.py; echo "res cap $cap: $(tail -1 $O/bs_res_$cap.log)"
    GPC_SPARSE_NO_RES=1 P=8192 CAP=$cap step 200 $O/bs_hbm_$cap.log python tools/bench_sparse.py; echo "hbm cap $cap: $(tail -1 $O/bs_hbm_$cap.log)"
  done
done
step 600 $O/pytest3b.log python -m pytest tests/test_sparse_gpu.py tests/test_probit_gpu.py -q -m gpu
tail -5 $O/pytest3b.log
