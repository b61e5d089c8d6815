#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
echo "none"; LIKE_BENCH=none timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "hwq8"; GPU_MAX_HW_QUEUES=8 LIKE_BENCH=none timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "prio -1"; GPC_C2_PRIO=-1 LIKE_BENCH=none timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "prio 1"; GPC_C2_PRIO=1 LIKE_BENCH=none timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "plain"; timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
echo "plain hwq2"; GPU_MAX_HW_QUEUES=2 timeout -k 10 120 python tools/r4_host_entry.py 2>/dev/null | tail -1 | cut -c1-110
