# One measurement pass for profiles/: default bench line, kernel-trace stats (default and single task), HBM-traffic PMC passes.
# Run on the GPU box from the repo root: bash tools/measure_round.sh ; then here: python tools/save_profiles.py rNN && python tools/pmc_traffic.py rNN <sha> && python tools/sq_counters.py rNN
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && echo "bench done" && tail -c 400 $O/bench_default.json
rm -rf $O/prof_k4 $O/prof_single $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
Q="--no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e"
rocprofv3 --kernel-trace --stats -d $O/prof_k4 -o k4 --output-format csv -- python3 bench.py --steps 30 --warmup 5 $Q > $O/prof_k4.log 2>&1 && echo "k4 profile done"
rocprofv3 --kernel-trace --stats -d $O/prof_single -o s --output-format csv -- python3 bench.py --steps 30 --warmup 5 $Q --tasks-per-gpu 1 > $O/prof_single.log 2>&1 && echo "single profile done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p --output-format csv -- python3 bench.py --steps 3 --warmup 2 $Q --tasks-per-gpu 1 > $O/pmc_$C.log 2>&1 && echo "pmc $C done"
done
rm -rf $O/pmc_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python3 bench.py --steps 3 --warmup 2 $Q --tasks-per-gpu 1 > $O/pmc_sq.log 2>&1 && echo "pmc sq done"
find $O/prof_k4 $O/prof_single -name "*kernel_trace.csv" -delete      # (the per-dispatch traces are large; the stats files are what is kept)
# BASELINE configs[0]: the BLSTM-CTC training step (CTC lattice + LSTM step kernels)
rm -rf $O/prof_blstm
rocprofv3 --kernel-trace --stats -d $O/prof_blstm -o b --output-format csv -- python3 tools/bench_blstm.py --steps 20 --warm 3 --out gpurun_out/blstm.json > $O/prof_blstm.log 2>&1 && echo "blstm profile done"
find $O/prof_blstm -name "*kernel_trace.csv" -delete
# effective shader clock per conv launch: GRBM_GUI_ACTIVE / 8 / wall on back-to-back dispatches of >= 1 ms (B = 256: the quotient reads high on short ones)
rm -rf $O/pmc_grbm_conv $O/pmc_grbm
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_grbm_conv -o p --output-format csv -- python3 tools/bench_conv.py 256 150 > $O/pmc_grbm_conv.log 2>&1 && echo "pmc grbm conv done"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_grbm -o p --output-format csv -- python3 bench.py --steps 40 --warmup 10 $Q --tasks-per-gpu 1 > $O/pmc_grbm.log 2>&1 && echo "pmc grbm step done"
find $O/pmc_grbm_conv $O/pmc_grbm -name "*kernel_trace.csv" -delete
python3 tools/grbm_clock.py $O/pmc_grbm_conv > $O/grbm_clock_conv.txt && python3 tools/grbm_clock.py $O/pmc_grbm > $O/grbm_clock_step.txt && head -8 $O/grbm_clock_conv.txt
