#!/bin/bash
# Round 4, GPU session 5: rocprofv3 evidence for every bench record (tools/profile_r04.sh), then the driver-shaped bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
bash tools/profile_r04.sh > $O/prof_all.log 2>&1; tail -4 $O/prof_all.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench5.json 2> $O/bench5.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench5.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:70], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"])
PY
