#!/bin/bash
# instruction mix of the sparse add kernels of a C4 bench record (default: "C4 defaults"; tools/profile_sparse_small.sh c4fill)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_sparse_small_${1:-c4defaults}
rm -rf $OUT; mkdir -p $OUT
W=${1:-c4defaults}
B="python3 bench.py --only $W --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/c -- $B > $OUT/c.log 2>&1
find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete
echo done
