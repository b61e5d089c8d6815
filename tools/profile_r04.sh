#!/bin/bash
# rocprofv3 evidence for profiles/ (round 4).  For the headline (C2) the kernel trace is taken from the SAME command the driver times
# (`bench.py --steps 20 --warmup 5`, without the CPU baseline and the secondary records: they run after the headline's timed region and
# do not touch it), so that the timed median of the trace and the line's own HIP-event median can be compared.
# Counters (--pmc) in their own passes with --kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950.
# Sparse workloads additionally get the SQ instruction-mix passes their issue-rate roofline is computed from.
# Outputs under gpurun_out/prof_r04/<workload>/<pass>/; tools/collect_profiles_r04.py derives profiles/r04_* from them.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
WL="${1:-c2 c2var c3 c4fill c4defaults c4defaults3 c4fills c4defaultss c5}"
mkdir -p $OUT
run() { local d=$1; shift; echo "=== $d: $*" | tee -a $OUT/session.log; rm -rf $OUT/$d; mkdir -p $OUT/$d
        timeout -k 10 300 "$@" > $OUT/$d/cmd.log 2>&1; rc=$?; echo "rc=$rc $(date +%T)" | tee -a $OUT/session.log
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $OUT/session.log; exit 1; fi; }
for w in $WL; do
  if [ $w = c2 ]; then B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"; C="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
  else B="python3 bench.py --only $w"; C="$B"; fi
  run $w/trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace/out -- $B
  run $w/fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$w/fetch/out -- $C
  run $w/write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$w/write/out -- $C
  case $w in
    c2|c2var|c3|c5)
      run $w/sq rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$w/sq/out -- $C
      run $w/grbm rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$w/grbm/out -- $C ;;
    c4fill|c4defaults|c4defaults3)
      run $w/sqa rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/$w/sqa/out -- $C
      run $w/sqb rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/$w/sqb/out -- $C
      run $w/sqc rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/$w/sqc/out -- $C ;;
    c4fills|c4defaultss)
      run $w/sq rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$w/sq/out -- $C ;;
  esac
done
# keep what travels back small: the per-dispatch CSVs only
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
echo done | tee -a $OUT/session.log
