cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
run() { python bench.py --no-secondary --no-cpu-baseline --steps 10 --points $2 --patches $3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 n=$2 P=$3', round(r['ms_per_step'],3), r['config']['kernel'], r['config']['results_ok'])"; }
for n in 64 128 192; do
  for P in 8192 512 64; do
    run reg $n $P
    GPC_W1_ALL=1 run w1 $n $P
  done
done
