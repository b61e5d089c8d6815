#!/bin/bash
# thread-safety of the two-stream host pipeline: the new test without the ordering (GPC_NO_PIPE_ORDER=1: expected to fail) and with it; then
# the whole GPU suite and the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
T=tests/test_dense_gpu.py::test_two_stream_pipeline_beside_another_threads_call
GPC_NO_PIPE_ORDER=1 timeout -k 10 300 python -m pytest $T -q -m gpu > $O/pytest36_off.log 2>&1; echo "ordering off rc=$? (1 expected)"; grep -n "^E " $O/pytest36_off.log | head -5 | cut -c1-200
timeout -k 10 300 python -m pytest $T -q -m gpu > $O/pytest36_on.log 2>&1; echo "ordering on rc=$?"; tail -3 $O/pytest36_on.log | cut -c1-200
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest36.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest36.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench36.json 2> $O/bench36.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench36.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"], r["host_pointer_entry"]["ms_per_call"], r["host_pointer_entry"]["results_equal_device_entry"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:86], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"], (s.get("roofline_predict") or {}).get("frac"))
PY
