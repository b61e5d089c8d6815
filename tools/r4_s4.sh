#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python tools/r4_overlap_probe.py > $O/overlap_probe.json 2> $O/overlap_probe.err; echo "probe rc=$?"; cat $O/overlap_probe.json
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest4.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest4.log
