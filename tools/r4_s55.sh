#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python tools/r4_stress_host.py ${1:-30} ${2:-1} > $O/stress_host.json 2> $O/stress_host.err; echo "rc=$?"; tail -3 $O/stress_host.err | cut -c1-200; cat $O/stress_host.json | cut -c1-400
