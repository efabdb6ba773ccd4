"""event-timed per-slot kernel times of one single-task inner step (the engine's profile slots): python tools/slot_times.py [slot ...]"""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-meta-step", "--long-seconds", "0", "--no-matrix",
                      "--no-mixed", "--no-e2e"], capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
want = sys.argv[1:] or list(k)
print({s: round(k[s] * 1e3, 1) for s in want}, "us;", "single", round(d["single_task_fomaml"]["value"]), "4-slot", round(d["value"]))
