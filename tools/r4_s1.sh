#!/bin/bash
# Round 4, GPU session 1: the suite, the driver-shaped bench line, the blow-up rate over 8 batches, the C2 profile at the bench's own steps,
# and the instruction / time shares of the headline kernel's phases (diagnostic builds, same box).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
step() { local secs=$1 log=$2; shift 2; echo "=== $* (limit ${secs}s) $(date +%T)" | tee -a $O/session1.log
         timeout -k 10 "$secs" "$@" > "$log" 2>&1; local rc=$?; echo "rc=$rc $(date +%T)" | tee -a $O/session1.log
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a $O/session1.log; exit 1; fi; return $rc; }
step 900 $O/pytest1.log python -m pytest tests -q -m gpu
tail -15 $O/pytest1.log
step 600 $O/bench1.err bash -c "python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench1.json"
python - <<PY
import json
r=json.load(open("$O/bench1.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:60], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"])
PY
step 600 $O/blowup.err bash -c "python tools/r4_blowup_rate.py > $O/blowup_rate.json"
tail -12 $O/blowup.err
bash tools/profile_r04.sh c2 > $O/prof_c2.log 2>&1; tail -3 $O/prof_c2.log
bash tools/r3_exp.sh nogram noback nopred nopasstrsm hot > $O/exp_time.log 2>&1; cat $O/exp_time.log
bash tools/r3_exp_pmc.sh base nogram noback nopred nopasstrsm > $O/exp_pmc.log 2>&1; tail -8 $O/exp_pmc.log
