"""MFMA utilisation / wait / LDS-conflict fractions per kernel of a single-task inner step, from one rocprofv3 PMC pass.

On the GPU box (repo root):
    cd /tmp && export TMPDIR=/tmp && cd - && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmc_sq -o p --output-format csv \
        -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e --tasks-per-gpu 1
then here:  python tools/sq_counters.py <round-tag>   ->  profiles/<tag>_sq_counters_single_task.txt
mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines); the other fractions are of SQ_WAVE_CYCLES."""
import csv
import glob
import sys
from collections import defaultdict

tag = sys.argv[1]
import os
f = max(glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)      # (older passes may still lie in gpurun_out/)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    wave = max(c["SQ_WAVE_CYCLES"], 1.0)
    busy = max(c["SQ_BUSY_CYCLES"], 1.0)
    rows.append((c["SQ_BUSY_CYCLES"], k, len(calls[k]), (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (busy / 32), c["SQ_WAIT_ANY"] / wave,
                 c["SQ_WAIT_INST_LDS"] / wave, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), c["SQ_ACTIVE_INST_ANY"] / wave))
rows.sort(reverse=True)
with open(f"profiles/{tag}_sq_counters_single_task.txt", "w") as out:
    out.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace\n")
    out.write("#   -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e --tasks-per-gpu 1   (MI355X; means per kernel, sorted by total busy time)\n")
    out.write("# mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines); the other fractions are of SQ_WAVE_CYCLES\n")
    out.write(f"{'kernel':84s} calls mfma_busy wait_any wait_lds lds_conflict issuing\n")
    for _, k, n, mf, wa, wl, lc, iss in rows:
        out.write(f"{k[:82]:84s} {n:5d} {mf:9.2f} {wa:8.2f} {wl:8.2f} {lc:12.2f} {iss:7.2f}\n")
print(open(f"profiles/{tag}_sq_counters_single_task.txt").read()[:3000])
