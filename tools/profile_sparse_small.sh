#!/bin/bash
# instruction mix of the small-basis sparse add kernel at the reference's default hyper-parameters (bench record "C4 defaults")
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_sparse_small
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --only c4defaults --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/c -- $B > $OUT/c.log 2>&1
find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete
echo done
