# full GPU suite, then the default bench line and a single-task per-launch timeline (GPU box, repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r5_gputests.log 2>&1 || { tail -30 $O/r5_gputests.log; exit 1; }
tail -3 $O/r5_gputests.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && python3 -c "
import json; d = json.load(open('$O/bench_default.json')); print('value', d['value'], 'single', d['single_task_fomaml']['value'], 'e2e_train', d.get('e2e_train', {}).get('sgd', d.get('e2e_train')), 'e2e_pretrain', d.get('e2e_pretrain', {}).get('value'))"
Q="--no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e"
rm -rf $O/trace_single
rocprofv3 --kernel-trace -d $O/trace_single -o s --output-format csv -- python3 bench.py --steps 12 --warmup 5 $Q --tasks-per-gpu 1 > $O/trace_single.log 2>&1
T=$(find $O/trace_single -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $T 3 > $O/timeline_single.txt && tail -1 $O/timeline_single.txt
find $O/trace_single -name "*kernel_trace.csv" -delete
