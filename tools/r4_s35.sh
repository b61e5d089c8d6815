#!/bin/bash
# two-stream host pipeline: (1) the new mixed-chunk test against the PREVIOUS gpc_api.hip (expected to fail: that is the bug), (2) against
# the fixed library, (3) the host-pointer entry timed from a torch side stream (bench) and the sigma-predict records
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
T=tests/test_dense_gpu.py::test_host_pointer_two_stream_mode_with_a_chunk_of_small_patches
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_oldapi.so timeout -k 10 300 python -m pytest $T -q -m gpu > $O/pytest35_old.log 2>&1; echo "old api rc=$? (1 expected)"; tail -5 $O/pytest35_old.log | cut -c1-200
timeout -k 10 300 python -m pytest $T -q -m gpu > $O/pytest35_new.log 2>&1; echo "new api rc=$?"; tail -3 $O/pytest35_new.log
timeout -k 10 600 python -m pytest tests/test_dense_gpu.py tests/test_sparse_gpu.py -q -m gpu > $O/pytest35.log 2>&1; echo "dense+sparse rc=$?"; tail -3 $O/pytest35.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench35.json 2> $O/bench35.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$O/bench35.json"))
print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms_stats"], r["host_pointer_entry"]["ms_per_call"], r["host_pointer_entry"]["results_equal_device_entry"])
for s in r.get("secondary",[]):
    print(s["config"]["workload"][:86], round(s["value"],1), s["roofline"]["bound"], round(s["roofline"]["frac"],4), s["config"]["results_ok"], (s.get("roofline_predict") or {}).get("frac"))
PY
