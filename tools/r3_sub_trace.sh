#!/bin/bash
# kernel timeline of one C4-defaults pass (start / end per dispatch): do the tails of the sub-phases overlap the rows launches?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3sub; mkdir -p $O
for v in ${@:-4 1}; do
  export GPC_SPARSE_NSUB=$v
  rm -rf $O/trace$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace$v -- python3 bench.py --only c4defaults --steps 2 > $O/trace$v.log 2>&1 || exit 1
  python3 - "$O/trace$v" <<'PY' | tee $O/timeline$v.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass: from the last-but-3 rows kernel group ... take the last 60 dispatches of sparse kernels
sp = [r for r in rows if "sparse" in r["Kernel_Name"]]
last = sp[-60:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"]
    short = "rows" if "rows" in n else "small" if "<true" in n else "regular" if "sparse_add" in n else "predict" if "predict" in n else n[:20]
    print(f'{short:8s} q={r.get("Queue_Id","?"):>3s} {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} -> {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us  ({(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f})  grid={r.get("Grid_Size","?")}')
PY
done
find $O -name "*.csv" -size +5M -delete
