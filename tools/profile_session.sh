#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then the HBM traffic counters in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; --pmc is never combined with trace domains other
# than --kernel-trace).  Outputs under gpurun_out/prof_<tag>/; tools/collect_profiles.py copies the summaries.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
run() { echo "=== $*" | tee -a $OUT/session.log; timeout -k 10 400 "$@" >> $OUT/session.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $OUT/session.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
run rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH
run rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH
run rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $BENCH
find $OUT -name "*.csv" | head -30
echo done | tee -a $OUT/session.log
