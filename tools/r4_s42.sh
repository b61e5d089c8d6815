#!/bin/bash
# per-kernel times of the C4-defaults pass after the rows-phase work (kernel trace), depth and colour
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for w in c4defaults c4defaults3; do
  rm -rf $O/kt_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 bench.py --only $w --no-cpu-baseline > $O/kt_$w.log 2>&1; echo "$w rc=$?"
  f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6-7 $f | head -8
  find $O/kt_$w -type f ! -name "*kernel_stats.csv" -delete
done
