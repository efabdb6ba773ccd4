"""Copy the rocprofv3 summaries of the last measurement run from gpurun_out/ into profiles/ (tracked).
usage: python tools/save_profiles.py <round-tag>   e.g. r01"""
import csv, json, shutil, sys
from pathlib import Path
tag = sys.argv[1]
out = Path("profiles"); g = Path("gpurun_out")


def summary(src, steps, header, dst):
    rows = list(csv.DictReader(open(src)))
    ks = []
    for r in rows:
        n = int(r["Calls"]); t = float(r["TotalDurationNs"])
        if n >= steps:
            ks.append((t / steps / 1e6, n / steps, t / n / 1e3, r["Name"]))
    ks.sort(reverse=True)
    tot = sum(k[0] for k in ks)
    with open(dst, "w") as f:
        f.write(f"# {header}\n# {steps} inner steps in the trace; total kernel time {tot:.3f} ms per inner step over {sum(k[1] for k in ks):.0f} launches\n")
        f.write(f"{'kernel':92s} calls/step   ms/step    avg_us    pct\n")
        for ms, c, avg, name in ks:
            f.write(f"{name[:90]:92s} {c:9.1f} {ms:9.3f} {avg:9.2f} {100 * ms / tot:6.2f}\n")


line = open(g / "bench_default.json").read().strip().splitlines()[-1]
json.loads(line)
(out / f"{tag}_bench.json").write_text(line + "\n")
import glob
k4 = glob.glob(str(g / "prof_k4/**/*kernel_stats.csv"), recursive=True)[0]
sg = glob.glob(str(g / "prof_single/**/*kernel_stats.csv"), recursive=True)[0]
shutil.copy(k4, out / f"{tag}_bench_kernel_stats.csv")
shutil.copy(sg, out / f"{tag}_single_task_kernel_stats.csv")
summary(k4, 35 * 4 + 15,
        "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e  (MI355X, default = 4 concurrent tasks; + the 10+5-step single-task leg)",
        out / f"{tag}_bench_kernel_stats.txt")
summary(sg, 35,
        "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-meta-step --long-seconds 0 --single-seconds 0 --no-matrix --no-mixed --no-e2e --tasks-per-gpu 1  (MI355X, one task per GPU)",
        out / f"{tag}_single_task_kernel_stats.txt")
bl = glob.glob(str(g / "prof_blstm/**/*kernel_stats.csv"), recursive=True)
if bl:
    shutil.copy(bl[0], out / f"{tag}_blstm_kernel_stats.csv")
    summary(bl[0], 23, "rocprofv3 --kernel-trace --stats -- python3 tools/bench_blstm.py --steps 20 --warm 3  (MI355X; BASELINE configs[0]: BLSTM-CTC training "
                       "step, B = 8 x 400 frames, 3 x BLSTM-P(360); the CTC lattice = ctc_lse / ctc_sweep / ctc_grad, the recurrence = lstm_fwd_rec / lstm_bwd_rec: one resident launch per layer and pass)",
            out / f"{tag}_blstm_kernel_stats.txt")
    if (g / "blstm.json").exists():
        shutil.copy(g / "blstm.json", out / f"{tag}_blstm.json")
print("saved")
