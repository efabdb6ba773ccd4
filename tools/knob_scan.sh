#!/bin/bash
# A/B of launch-geometry knobs on the default (4 task slots) and the one-task leg of bench.py, one box, back to back.
# usage: bash tools/knob_scan.sh "NAME=VALUE ..." "NAME=VALUE ..."   (each argument = one environment; "" = the shipped defaults)
mkdir -p gpurun_out
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 2 --no-matrix --no-mixed --no-e2e"
for cfg in "$@"; do
  out=$(env $cfg python3 bench.py $ARGS 2>/dev/null)
  echo "$cfg :: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("4-slot %.0f (long %.0f) single %.0f" % (d["value"], d["long_run"]["value"], d["single_task_fomaml"]["value"]))')"
done
