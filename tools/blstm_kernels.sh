# per-kernel times of the BLSTM-CTC training step (GPU box, repo root): bash tools/blstm_kernels.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_blstm_$1
rm -rf $O
rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python3 tools/bench_blstm.py --steps 10 > $O.log 2>&1
python3 - "$O/b_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.3f ms (13 steps)" % (tot / 13 / 1e6))
for r in rows[:14]:
    print("%8.1f us x %5.1f /step  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]) / 13, r["Name"][:110]))
PY
tail -1 $O.log
find $O -name "*kernel_trace.csv" -delete
