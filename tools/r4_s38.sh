#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 500 python tools/r4_tail_probe.py > $O/tail_probe.json 2> $O/tail_probe.err; echo "rc=$?"; tail -3 $O/tail_probe.err; cat $O/tail_probe.json
