#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest20.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest20.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench20.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"], r["host_pointer_entry"]["ms_per_call"], r["host_pointer_entry"]["results_equal_device_entry"], r["roofline"]["kernel_ms_rocprof"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:86], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"], (s.get("roofline_predict") or {}).get("frac"))
PY
