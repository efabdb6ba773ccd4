#!/bin/bash
# how much of a single task's exposed decoder latency would hiding it behind another micro-batch of the SAME task buy?  Upper bound:
# K concurrent slots of B/K utterances each (independent tasks here; a split task adds one pass that sums the K gradient buffers).
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 2 --no-matrix --no-mixed --no-e2e"
for cfg in "1 16" "2 8" "4 4" "2 16" "3 16"; do
  set -- $cfg
  out=$(python3 bench.py $ARGS --tasks-per-gpu $1 --batch $2 2>/dev/null)
  echo "slots $1 x B $2 :: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f utt/s (long %.0f), %.3f ms per round" % (d["value"], d["long_run"]["value"], d["long_run"]["ms_per_step"]))')"
done
