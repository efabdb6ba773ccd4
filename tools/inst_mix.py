"""Instruction mix and modelled energy per kernel of a single-task inner step, from one rocprofv3 PMC pass (wave-instruction counts).

On the GPU box (repo root):
    cd /tmp && export TMPDIR=/tmp && cd - && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES \
        --kernel-trace -d gpurun_out/pmc_inst -o p --output-format csv \
        -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e --tasks-per-gpu 1
then here:  python tools/inst_mix.py <round-tag>   ->  profiles/<tag>_inst_mix_single_task.txt
Energy model (tools/energy/energy_probe.hip): 9.5 nJ per bf16 MFMA, 1.35 nJ per other vector instruction, 4.1 nJ per LDS instruction
(priced as a 1 KB read: an upper bound for the narrower ones).  SQ_INSTS_VALU counts the MFMAs too (subtracted below)."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
f = max(glob.glob("gpurun_out/pmc_inst/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
steps = 5                                                        # --steps 3 --warmup 2
rows = []
for k, c in acc.items():
    n = len(calls[k])
    if n < steps:
        continue
    mf = c["SQ_INSTS_MFMA"] / steps
    va = max(c["SQ_INSTS_VALU"] / steps - mf, 0.0)
    ld = c["SQ_INSTS_LDS"] / steps
    sa, vr, vw = c["SQ_INSTS_SALU"] / steps, c["SQ_INSTS_VMEM_RD"] / steps, c["SQ_INSTS_VMEM_WR"] / steps
    e = (mf * 9.5 + va * 1.35 + ld * 4.1) * 1e-6                 # mJ per step
    rows.append((e, k, n / steps, mf, va, ld, sa, vr + vw))
rows.sort(reverse=True)
tot = [sum(r[i] for r in rows) for i in (0, 3, 4, 5, 6, 7)]
with open(f"profiles/{tag}_inst_mix_single_task.txt", "w") as out:
    out.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace -- python3 bench.py --steps 3 --warmup 2 ... --tasks-per-gpu 1\n")
    out.write("# wave-instructions per inner step (millions) and modelled energy: 9.5 nJ per MFMA + 1.35 nJ per other vector instruction + 4.1 nJ per LDS instruction\n")
    out.write(f"# step total: {tot[1] / 1e6:.1f} M MFMA, {tot[2] / 1e6:.1f} M vector, {tot[3] / 1e6:.1f} M LDS, {tot[4] / 1e6:.1f} M scalar, {tot[5] / 1e6:.1f} M vector-memory: "
              f"{tot[1] * 9.5e-6:.0f} + {tot[2] * 1.35e-6:.0f} + {tot[3] * 4.1e-6:.0f} = {tot[0]:.0f} mJ of instruction energy per step\n")
    out.write(f"{'kernel':78s} calls   MFMA  vector    LDS scalar   vmem  vec/MFMA LDS/MFMA  mJ/step\n")
    for e, k, n, mf, va, ld, sa, vm in rows:
        out.write(f"{k[:76]:78s} {n:5.1f} {mf / 1e6:6.2f} {va / 1e6:7.2f} {ld / 1e6:6.2f} {sa / 1e6:6.2f} {vm / 1e6:6.2f} {va / mf if mf else 0:9.2f} {ld / mf if mf else 0:8.2f} {e:8.2f}\n")
print(open(f"profiles/{tag}_inst_mix_single_task.txt").read()[:4500])
