#!/bin/bash
# end of round 4: build from the sources as committed, the whole GPU suite, smoke(), the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest60.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest60.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/bench60.json 2> $O/bench60.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench60.json"))
print("headline", round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["steps"], r["warmup"], r["host_pointer_entry"]["ms_per_call"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:80], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"])
PY
