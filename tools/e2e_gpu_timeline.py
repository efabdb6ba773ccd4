"""GPU-side timeline of the K-tasks-per-GPU pretrain loop WITHOUT a profiler (under rocprofv3 the host is slowed enough to change which
side is the bottleneck): timing events on the task streams (released / inner step done / val batch + clip done) and on the main stream
(joins passed / meta update done), printed relative to the end of the previous meta update.  This is what showed that a fifth busy HIP
stream costs a whole task's worth of time: three tasks done after 14.8 ms, the fourth after 23.3 ms.

    python tools/e2e_gpu_timeline.py [--configs host:4] [--sync-stats]        (arguments of tools/bench_pretrain.py)"""
import os, sys, time, statistics, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
args = sys.argv[1:]
sys.argv = ["bench_pretrain.py", "--utts", "2048", "--meta-steps", "40"] + (args or ["--configs", "host:4"])
sys.path.insert(0, ".")
import torch
import masr_amd
from masr_amd.fo_meta_interface import FOMetaASRInterface as F
T = time.perf_counter
E = lambda: torch.cuda.Event(enable_timing=True)
log = []          # per meta-step: dict
cur = {}
o_task = F._task_on_slot
def task(self, slot, tr, val, out, i):
    with torch.cuda.stream(slot['stream']):
        e = E(); e.record(); cur.setdefault("start", {})[i] = e
    r = o_task(self, slot, tr, val, out, i)
    with torch.cuda.stream(slot['stream']):
        e = E(); e.record(); cur.setdefault("end", {})[i] = e
    return r
F._task_on_slot = task
o_rt = F.run_task
def run_task(self, b, engine=None):
    r = o_rt(self, b, engine=engine)
    e = E(); e.record(); cur.setdefault("inner", []).append(e)
    return r
F.run_task = run_task
o_fin = F._final_meta_update
def fin(self, n=None):
    global cur
    e = E(); e.record(); cur["join"] = e          # main stream: after the waits on the slots, before the accumulations' scale + Adam
    r = o_fin(self, n)
    e = E(); e.record(); cur["adam"] = e; cur["host"] = T()
    log.append(cur); cur = {}
    return r
F._final_meta_update = fin
import runpy
runpy.run_path("tools/bench_pretrain.py", run_name="__main__")
torch.cuda.synchronize()
rows = []
for k in range(12, len(log) - 1):
    p, c = log[k - 1], log[k]
    o = p["adam"]
    rows.append({"starts": sorted(o.elapsed_time(e) for e in c["start"].values()), "inner": sorted(o.elapsed_time(e) for e in c["inner"]),
                 "ends": sorted(o.elapsed_time(e) for e in c["end"].values()), "join": o.elapsed_time(c["join"]), "adam": o.elapsed_time(c["adam"])})
def med(f): return [round(statistics.median(x), 2) for x in zip(*[f(r) for r in rows])]
print("GPU ms after the previous meta update finished (medians over %d meta-steps):" % len(rows), file=sys.stderr)
print("  task streams released      ", med(lambda r: r["starts"]), file=sys.stderr)
print("  inner steps done           ", med(lambda r: r["inner"]), file=sys.stderr)
print("  val batch + clip done      ", med(lambda r: r["ends"]), file=sys.stderr)
print("  main stream past the joins ", round(statistics.median(r["join"] for r in rows), 2), " meta update done", round(statistics.median(r["adam"] for r in rows), 2), file=sys.stderr)
