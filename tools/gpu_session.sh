#!/bin/bash
# One gpurun call: parity tests, MFMA probe, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# A step that is killed by its timeout stops the session (no further GPU work after a hang).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
TAG=${1:-s}
step() {  # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "=== $* (limit ${secs}s)" | tee -a $OUT/session_$TAG.log
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/session_$TAG.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping session" | tee -a $OUT/session_$TAG.log; exit 1; fi
    return $rc
}
if [ "${RUN_PROBE:-1}" = "1" ]; then
  hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_f64.hip -o /tmp/probe_mfma > $OUT/probe_build_$TAG.log 2>&1
  step 120 $OUT/probe_$TAG.log /tmp/probe_mfma
fi
if [ "${RUN_TESTS:-1}" = "1" ]; then
  step ${TEST_LIMIT:-900} $OUT/pytest_$TAG.log python -m pytest tests -q -m gpu ${PYTEST_ARGS:-}
  tail -5 $OUT/pytest_$TAG.log
fi
if [ "${RUN_STAMP:-0}" = "1" ]; then
  step 300 $OUT/stamp_$TAG.log python tools/stamp_mfma.py
  tail -16 $OUT/stamp_$TAG.log
  if [ -f gp_compressor_amd/libgpc_hip_stamps2.so ]; then
    GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_stamps2.so step 300 $OUT/stamp2_$TAG.log python tools/stamp_mfma.py
  fi
fi
if [ "${RUN_BENCH:-1}" = "1" ]; then
  step 600 $OUT/bench_$TAG.log python bench.py --steps ${BENCH_STEPS:-10} --warmup 2
  tail -2 $OUT/bench_$TAG.log
fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  rm -rf $OUT/prof_$TAG
  step 600 $OUT/rocprof_$TAG.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
  find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -r head -20
fi
echo "session done" | tee -a $OUT/session_$TAG.log
