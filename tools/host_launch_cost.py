"""How long does the HOST take to enqueue one inner step (run_batch + clip_sgd: ~170 launches), alone and with K threads
enqueueing at once?  If K x that time approaches the K-task step time, the concurrent-task mode is launch-bound and a
captured graph per step would raise the throughput; if not, the GPU is the limit."""
import os, sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from masr_amd.engine import MasrEngine
from masr_amd.model import reference_init_state_dict

cfg = dict(bench.HKUST); dev = torch.device("cuda:0")
torch.manual_seed(531); sd = reference_init_state_dict(cfg, bench.ODIM)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tasks = []
for k in range(K):
    e = MasrEngine(cfg, bench.ODIM, 0.2, dev); e.load_state_dict(sd)
    e.set_step_graphs(os.environ.get("GRAPH", "0") == "1")
    xs, il, ys, ol = bench.synth_batch(16, 1000, 80, k)
    tasks.append(dict(e=e, xs=xs.to(dev), il=il, ys=ys, ol=ol, mom=torch.zeros_like(e.params), s=torch.cuda.Stream(dev)))

def step(t, first=False):
    t["e"].run_batch(t["xs"], t["il"], t["ys"], t["ol"], train=True)
    t["e"].clip_sgd_step(t["mom"], 5.0, 2.8e-4, 0.9, True, first)

for t in tasks:
    with torch.cuda.stream(t["s"]):
        for i in range(3): step(t, i == 0)
torch.cuda.synchronize()
# one thread, one task: enqueue N steps without waiting, time the enqueue alone, then the drain
N = int(os.environ.get('N_STEPS', '3'))          # few steps: beyond the depth of the hardware queue the enqueue blocks on the GPU
with torch.cuda.stream(tasks[0]["s"]):
    t0 = time.perf_counter()
    for _ in range(N): step(tasks[0])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"1 thread : enqueue {(t1 - t0) / N * 1e3:.3f} ms/step, GPU {(t2 - t0) / N * 1e3:.3f} ms/step")
res = [None] * K
gate = threading.Barrier(K)
def body(k):
    t = tasks[k]
    with torch.cuda.stream(t["s"]):
        gate.wait()
        a = time.perf_counter()
        for _ in range(N): step(t)
        b = time.perf_counter()
        t["s"].synchronize()
        res[k] = (b - a, time.perf_counter() - a)
ths = [threading.Thread(target=body, args=(k,)) for k in range(K)]
[th.start() for th in ths]; [th.join() for th in ths]
print(f"{K} threads: enqueue {max(r[0] for r in res) / N * 1e3:.3f} ms/step per thread, wall {max(r[1] for r in res) / N * 1e3:.3f} ms per round of {K} steps")
print("step counters of task 0:", tasks[0]["e"].step_counters())
