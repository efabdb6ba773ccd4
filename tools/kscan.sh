# usage: bash tools/kscan.sh "K list" "env settings ..."   -- bench.py throughput for each K under each environment
KS=${1:-"1 2 3 4 6"}; shift
for K in $KS; do
  for E in "$@"; do
    v=$(env $E timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile --tasks-per-gpu $K 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
    echo "K=$K [$E]: $v"
  done
done
