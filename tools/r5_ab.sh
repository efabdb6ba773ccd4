# quick A/B on one box: conv slot times of the single-task step (engine profile slots), 3 repeats
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-meta-step --long-seconds 0 --single-seconds 0.5 --no-matrix --no-mixed --no-e2e 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('value %.0f single %.0f' % (d['value'], d['single_task_fomaml']['value']), ' '.join('%s %.1f' % (n, k[n]*1e3) for n in ('conv2_dgrad','conv4_dgrad','conv2_fwd','conv1_wgrad','shadows','optim')))"; done
