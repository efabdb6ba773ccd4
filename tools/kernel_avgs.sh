# per-kernel average durations of the single-task step (rocprofv3 --kernel-trace --stats), top N lines: bash tools/kernel_avgs.sh [N]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/ka
rocprofv3 --kernel-trace --stats -d gpurun_out/ka -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e --tasks-per-gpu 1 > gpurun_out/ka.log 2>&1
python3 tools/kstats.py $(find gpurun_out/ka -name "*kernel_stats.csv") 25 ${1:-45} > gpurun_out/kernel_avgs.txt
python3 tools/step_timeline.py $(find gpurun_out/ka -name "*kernel_trace.csv") > gpurun_out/timeline.txt
find gpurun_out/ka -name "*.csv" -delete
