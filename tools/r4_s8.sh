#!/bin/bash
# Round 4, GPU session 8: the shipped library after the last kernel changes -- whole GPU suite, profiles of the workloads whose kernels
# changed (c2: zero-spill one-wave kernel; c4defaults / c4defaultss / c4defaults3: small-basis predict kernel), the driver-shaped bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest8.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest8.log
bash tools/profile_r04.sh "c2 c4defaults c4defaultss c4defaults3 c4fills" > $O/prof8.log 2>&1; tail -3 $O/prof8.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench8.json 2> $O/bench8.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench8.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"], r["host_pointer_entry"]["ms_per_call"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:70], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"])
PY
