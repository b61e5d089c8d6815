#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for m in dev page none 1; do
  echo "LIKE_BENCH=$m"; LIKE_BENCH=$m timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
done
echo "slots 1024"; LIKE_BENCH=dev GPC_W1_SLOTS=1024 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
