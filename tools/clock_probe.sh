# samples the shader / memory clocks and the board power while the default bench's long leg runs: bash tools/clock_probe.sh
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 8 --no-matrix --no-mixed --no-e2e > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 4
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
echo "idle:"; sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo
