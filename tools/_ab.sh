Q="--steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --no-matrix --no-mixed --no-e2e"
for e in "X=1" "MASR_WGRAD_WGS=512 MASR_WGRAD_OCC=2"; do
  for k in 1 4; do
    v=$(env $e python3 bench.py $Q --tasks-per-gpu $k 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))")
    echo "[$e] k=$k: $v"
  done
done
