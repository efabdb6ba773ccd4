# inner-step throughput against the number of concurrent task slots (GPU box, repo root): bash tools/slots_probe.sh
Q="--no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e"
for k in 2 3 4 5 6 8; do
  python3 bench.py --steps 30 --warmup 6 $Q --tasks-per-gpu $k 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $k value %.0f utt/s  %.3f ms per step' % (d['value'], d['ms_per_step']))"
done
