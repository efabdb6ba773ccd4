# Round-5 first look (GPU box, repo root): per-launch timeline of ONE single-task inner step, and the GRBM_GUI_ACTIVE pass that
# prices the clock each conv launch runs at (effective clock = GRBM_GUI_ACTIVE / 8 XCDs / wall; MI355X_MICROARCH.md 'DVFS give-back').
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
Q="--no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e"
rm -rf $O/trace_single $O/pmc_grbm $O/pmc_grbm_conv
rocprofv3 --kernel-trace -d $O/trace_single -o s --output-format csv -- python3 bench.py --steps 12 --warmup 5 $Q --tasks-per-gpu 1 > $O/trace_single.log 2>&1 && echo "trace done"
T=$(find $O/trace_single -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $T 3 > $O/timeline_single.txt && tail -3 $O/timeline_single.txt
find $O/trace_single -name "*kernel_trace.csv" -delete
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_grbm -o p --output-format csv -- python3 bench.py --steps 40 --warmup 10 $Q --tasks-per-gpu 1 > $O/pmc_grbm.log 2>&1 && echo "pmc grbm done"
find $O/pmc_grbm -name "*kernel_trace.csv" -delete
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_grbm_conv -o p --output-format csv -- python3 tools/bench_conv.py 64 400 > $O/pmc_grbm_conv.log 2>&1 && echo "pmc grbm conv done"
find $O/pmc_grbm_conv -name "*kernel_trace.csv" -delete
python3 tools/grbm_clock.py $O/pmc_grbm > $O/grbm_clock_step.txt && python3 tools/grbm_clock.py $O/pmc_grbm_conv > $O/grbm_clock_conv64.txt && cat $O/grbm_clock_conv64.txt
