#!/bin/bash
# Run on the GPU box: bash tools/energy/run.sh  -> per mode: instruction rates and the board power (rocm-smi --showpower, 6 samples from 1.5 s into a 6 s run)
set -e
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/energy_probe tools/energy/energy_probe.hip
pw() { rocm-smi --showpower | sed -n 's/.*Package Power (W): *\([0-9.]*\).*/\1/p' | head -1; }      # (the visible GPU; the hwmon files are per card of the host)
echo "idle power: $(pw) W"
for mode in ${MODES:-0 1 2 3 4 5 0}; do
  timeout -k 5 30 /tmp/energy_probe $mode 6 > /tmp/energy_$mode.txt &
  pid=$!
  sleep 1.5; samples=""
  for k in $(seq 6); do samples="$samples $(pw)"; sleep 0.3; done
  wait $pid
  echo "$(cat /tmp/energy_$mode.txt)   board power (W):$samples"
done
