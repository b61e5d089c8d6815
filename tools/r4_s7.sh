#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_sp2.so timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py -q -m gpu -k "small_basis_predict or predict_points or batch_vs_oracle" > $O/pytest7.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest7.log
GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_v3.so timeout -k 10 600 python -m pytest tests/test_dense_gpu.py -q -m gpu > $O/pytest7b.log 2>&1; echo "pytest v3 rc=$?"; tail -3 $O/pytest7b.log
bash tools/r3_exp.sh v3 r3w1 > $O/exp7_time.log 2>&1; cat $O/exp7_time.log
for v in base v3; do
  if [ $v = base ]; then unset GPC_LIB_PATH; else export GPC_LIB_PATH=$PWD/gp_compressor_amd/libgpc_hip_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc7_$v_$c; timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc7_${v}_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc7_${v}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob
for v in ("base","v3"):
    tot={}
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        vals=[]
        for f in glob.glob(f"gpurun_out/r4/pmc7_{v}_{c}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "dense_w1_kernel" in r["Kernel_Name"] and r["Counter_Name"]==c: vals.append(float(r["Counter_Value"]))
        tot[c]=sum(vals)/max(1,len(vals))
    print(v, "read GB", 2*tot["FETCH_SIZE"]*1024/1e9, "write GB", tot["WRITE_SIZE"]*1024/1e9)
PY
find $O -name "*.csv" -size +5M -delete
