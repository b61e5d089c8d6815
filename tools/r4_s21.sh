#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2; do
  NULL_STREAM=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-150
  NULL_STREAM=1 LIKE_BENCH=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-150
  LIKE_BENCH=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-150
  NULL_STREAM=1 LIKE_BENCH=1 GPC_HOST_ONE_STREAM=1 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-150
done
