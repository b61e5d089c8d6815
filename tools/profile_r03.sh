#!/bin/bash
# rocprofv3 evidence for profiles/ (round 3): for the headline (C2) and every secondary bench record (C3, C4 fill / defaults,
# C5) a kernel trace (per-dispatch durations -> the average over TIMED launches only) and the HBM traffic counters in their
# own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; --pmc only ever together with --kernel-trace).
# Outputs under gpurun_out/prof_r03/<workload>/<pass>/; tools/collect_profiles_r03.py derives profiles/r03_* from them.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
WL="${1:-c2 c2var c3 c4fill c4defaults c4defaults3 c5}"
mkdir -p $OUT
run() { local d=$1; shift; echo "=== $d: $*" | tee -a $OUT/session.log; rm -rf $OUT/$d; mkdir -p $OUT/$d
        timeout -k 10 300 "$@" > $OUT/$d/cmd.log 2>&1; rc=$?; echo "rc=$rc $(date +%T)" | tee -a $OUT/session.log
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $OUT/session.log; exit 1; fi; }
for w in $WL; do
  if [ $w = c2 ]; then B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
  else B="python3 bench.py --only $w"; fi
  run $w/trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace/out -- $B
  run $w/fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$w/fetch/out -- $B
  run $w/write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$w/write/out -- $B
  if [ $w = c2 ] || [ $w = c2var ] || [ $w = c3 ] || [ $w = c5 ]; then
    run $w/sq rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$w/sq/out -- $B
    run $w/grbm rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$w/grbm/out -- $B
  fi
done
# keep what travels back small: the per-dispatch CSVs only
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
echo done | tee -a $OUT/session.log
