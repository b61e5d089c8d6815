#!/bin/bash
# final checks of the round: smoke(), and the N-rank path rehearsed with ONE rank (full line with the secondary records: the sharded sparse
# record goes through the chained work lists too)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
GPC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_dist1_final.json 2> $O/bench_dist1_final.err; echo "dist1 rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench_dist1_final.json"))
print("headline", round(r["value"]), r["ms_per_step"], r["roofline"]["frac"], r["config"].get("results_ok"), r["scaling"], r["n_gpus"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:90], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"].get("results_ok"))
PY
