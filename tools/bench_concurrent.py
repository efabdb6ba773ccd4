"""Experiment: K independent tasks (engines) per GPU, each on its own HIP stream, one host thread per task
(ctypes releases the GIL inside libmasr calls, so launch overhead spreads over cores)."""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import masr_amd
from masr_amd.engine import MasrEngine
from oracle import ref_cpu
from bench import HKUST, ODIM, synth_batch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
threaded = len(sys.argv) > 2 and sys.argv[2] == "t"
B, T, D = 16, 1000, 80
cfg = dict(HKUST)
engs, streams, xs, moms = [], [], [], []
sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=1)
for k in range(K):
    e = MasrEngine(cfg, ODIM, label_smoothing=0.2)
    e.load_state_dict(sd); e.set_seed(531 + k)
    engs.append(e); streams.append(torch.cuda.Stream())
    b = synth_batch(B, T, D, seed=k)
    xs.append((b[0].cuda(), b[1], b[2], b[3])); moms.append(torch.zeros_like(e.params))
lr = ref_cpu.inner_lr(cfg)
def one(k, i):
    with torch.cuda.stream(streams[k]):
        engs[k].run_batch(*xs[k], train=True)
        engs[k].clip_sgd_step(moms[k], 5.0, lr, 0.9, True, first_step=(i == 0))
n = 30
def worker(k, bar):
    for i in range(5): one(k, i)
    torch.cuda.synchronize()
    bar.wait()
    for i in range(n): one(k, 5 + i)
    streams[k].synchronize()
if threaded:
    bar = threading.Barrier(K + 1)
    ths = [threading.Thread(target=worker, args=(k, bar)) for k in range(K)]
    for t in ths: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
else:
    for i in range(5):
        for k in range(K): one(k, i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        for k in range(K): one(k, 5 + i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"K={K} threaded={threaded}: {K*B*n/dt:.1f} utt/s, {dt/n*1e3:.3f} ms per round of {K} inner steps; loss {[round(e.read_stats()['loss'],4) for e in engs]}")
