#!/bin/bash
# instruction mix of the likelihood kernel (row f1) on the trained C4 state (tools/bench_sparse.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_sparse_lik
rm -rf $OUT; mkdir -p $OUT
B="python3 tools/bench_sparse.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- $B > $OUT/t.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
find $OUT -type f ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" ! -name "*.log" -delete
echo done
