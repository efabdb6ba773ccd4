# L2 hit / miss counts per kernel of the single-task step (rocprofv3 PMC pass): bash tools/pmc_l2.sh [kernel-name-substring ...]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
Q="--no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e"
rm -rf $O/pmc_l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $O/pmc_l2 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 2 $Q --tasks-per-gpu 1 > $O/pmc_l2.log 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
from collections import defaultdict
f = sorted(glob.glob("gpurun_out/pmc_l2/**/*counter_collection.csv", recursive=True))[-1]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pats = sys.argv[1:] or ["wgrad", "gemm_glds_kernel<128", "conv3x3"]
for k, v in sorted(acc.items()):
    if not any(p in k for p in pats):
        continue
    h, m = sum(v["TCC_HIT_sum"]) / len(v["TCC_HIT_sum"]), sum(v["TCC_MISS_sum"]) / len(v["TCC_MISS_sum"])
    rq = sum(v.get("TCC_REQ_sum", [0])) / max(len(v.get("TCC_REQ_sum", [0])), 1)
    print(f"{k[:110]:110s} hit {h:12.0f} miss {m:12.0f} req {rq:12.0f} hit-rate {h / (h + m + 1e-9):.3f}  launches {len(v['TCC_HIT_sum'])}")
PY
