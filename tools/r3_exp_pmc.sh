#!/bin/bash
# PMC passes over the diagnostic builds (see tools/r3_exp.sh): instruction mix and busy / wait cycles of the headline kernel per variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/exp_pmc; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for v in ${@:-base nogram noback nopred}; do
  if [ $v = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/$v/a -- $B > $O/$v.a.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/$v/b -- $B > $O/$v.b.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_SALU --output-format csv -d $O/$v/c -- $B > $O/$v.c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
O="gpurun_out/exp_pmc"
for v in sorted(os.listdir(O)):
    if not os.path.isdir(os.path.join(O,v)): continue
    acc=collections.defaultdict(list)
    for f in glob.glob(os.path.join(O,v,"**","*_counter_collection.csv"),recursive=True):
        for r in csv.DictReader(open(f)):
            if "dense_big_kernel" in r["Kernel_Name"] or "dense_w1_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x)/len(x)/8192,1) for k,x in sorted(acc.items())})
PY
find $O -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find $O -name "*.csv" -size +5M -delete
