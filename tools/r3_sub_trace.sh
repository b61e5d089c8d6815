#!/bin/bash
# kernel timeline of one pass of a sparse C4 record (start / end per dispatch from rocprofv3's kernel trace): which kernels overlap,
# where the gaps are.  Usage: tools/r3_sub_trace.sh [c4defaults|c4fill|c4defaults3]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3trace; mkdir -p $O
W=${1:-c4defaults}
rm -rf $O/trace_$W
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$W -- python3 bench.py --only $W --steps 2 > $O/trace_$W.log 2>&1 || exit 1
python3 - "$O/trace_$W" <<'PY' | tee $O/timeline_$W.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sp = [r for r in rows if "sparse" in r["Kernel_Name"]]
last = sp[-16:]                      # the last pass: 4 add calls x (rows, small-basis, regular) + predict, give or take
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"]
    short = "rows" if "rows" in n else "small" if "<true" in n else "regular" if "sparse_add" in n else "predict" if "predict" in n else n[:20]
    print(f'{short:8s} q={r.get("Queue_Id","?"):>3s} {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} -> {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us  ({(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f})')
PY
find $O -name "*.csv" -size +5M -delete
