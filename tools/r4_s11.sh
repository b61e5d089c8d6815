#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
GPC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist rc=$?"; tail -3 $O/bench_dist1.err
python - <<PY
import json
r=json.load(open("$O/bench_dist1.json"))
print("headline", r["value"], r["config"].get("exchange"), r["config"]["results_ok"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:80], round(s["value"],1), (s.get("roofline") or {}).get("bound"), round((s.get("roofline") or {}).get("frac",0),4), s["config"]["results_ok"])
PY
GPC_C4_STEPS=2 GPC_C4_CPU_S=3 timeout -k 10 300 python bench.py --only c4fill100 > $O/c4fill100.json 2> $O/c4fill100.err; echo "cap100 rc=$?"
python - <<PY
import json
r=json.load(open("$O/c4fill100.json"))
print(r["config"]["workload"][:90], round(r["value"],1), r["roofline"]["frac"], r["config"]["results_ok"], r.get("speedup_vs_cpu_baseline"), r["config"]["bv_mean"])
PY
