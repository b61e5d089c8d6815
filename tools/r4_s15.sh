#!/bin/bash
# Round 4, final GPU session: whole suite on the shipped library, profiles of the workloads whose kernels changed last (c2: host pipeline
# does not touch the device entry but the library was rebuilt; c4fills: deeper operand prefetch in sp_ck_chunk), the driver-shaped bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest15.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest15.log
bash tools/profile_r04.sh "c2 c4fills c4fill" > $O/prof15.log 2>&1; tail -3 $O/prof15.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench15.json 2> $O/bench15.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench15.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"], r["host_pointer_entry"]["ms_per_call"], r["host_pointer_entry"]["results_equal_device_entry"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:86], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"], (s.get("roofline_predict") or {}).get("frac"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
