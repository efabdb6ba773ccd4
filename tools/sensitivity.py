#!/usr/bin/env python3
"""What the 4-concurrent-task inner step is sensitive to: the same measurement as bench.py's headline leg (K task slots, one host thread
+ HIP stream each, slot 0 on the default stream) on variations of the model -- fewer decoder / encoder layers, no FFN to speak of,
shorter input -- so that the difference to the full model prices a part of the step UNDER CONCURRENCY (where latency-bound launches
ride along and only chip-filling work counts), which the single-task kernel sums cannot.
    python tools/sensitivity.py [--tasks 4] [--steps 30]"""
import argparse
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import masr_amd  # noqa: E402,F401
from masr_amd.engine import MasrEngine  # noqa: E402
from masr_amd.model import reference_init_state_dict  # noqa: E402
from bench import HKUST, ODIM, synth_batch  # noqa: E402


def run(cfg, B, T, K, steps, warm=5):
    dev = torch.device("cuda:0")
    torch.manual_seed(531)
    sd = reference_init_state_dict(cfg, ODIM)
    tasks = []
    for k in range(K):
        e = MasrEngine(cfg, ODIM, label_smoothing=0.2, device=dev)
        e.load_state_dict(sd); e.set_seed(531 + k); e.set_concurrency(K)
        xs, il, ys, ol = synth_batch(B, T, cfg["idim"], seed=k)
        tasks.append(dict(e=e, xs=xs.to(dev), il=il, ys=ys, ol=ol, mom=torch.zeros_like(e.params),
                          st=torch.cuda.current_stream(dev) if k == 0 else STREAMS[k - 1]))
    gate = threading.Barrier(K + 1)

    def body(t):
        with torch.cuda.stream(t["st"]):
            for i in range(warm):
                t["e"].run_batch(t["xs"], t["il"], t["ys"], t["ol"], train=True); t["e"].clip_sgd_step(t["mom"], 5.0, 2.8e-4, 0.9, True, i == 0)
            t["st"].synchronize(); gate.wait(); gate.wait()
            for i in range(steps):
                t["e"].run_batch(t["xs"], t["il"], t["ys"], t["ol"], train=True); t["e"].clip_sgd_step(t["mom"], 5.0, 2.8e-4, 0.9, True, False)
            t["st"].synchronize()
    ths = [threading.Thread(target=body, args=(t,)) for t in tasks]
    for th in ths: th.start()
    gate.wait(); torch.cuda.synchronize(); t0 = time.perf_counter(); gate.wait()
    for th in ths: th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del tasks
    torch.cuda.empty_cache()
    return dt / steps * 1e3


STREAMS = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    STREAMS.extend(torch.cuda.Stream() for _ in range(a.tasks - 1))
    base = dict(HKUST)
    variants = [("full model", {}, 1000),
                ("decoder 4 -> 1 layers", {"decoder": {"nlayers": 1}}, 1000),
                ("encoder 2 -> 1 layers", {"encoder": {"nlayers": 1}}, 1000),
                ("d_inner 2048 -> 64 (FFN GEMMs gone)", {"d_inner": 64}, 1000),
                ("frames 1000 -> 500 (convs, encoder rows halved)", {}, 500),
                ("full model again", {}, 1000)]
    ref = None
    for name, over, T in variants:
        for K in (1, a.tasks):
            ms = run(dict(base, **over), a.batch, T, K, a.steps)
            if name == "full model":
                ref = ref or {}
                ref[K] = ms
            print(f"{name:52s} K={K}: {ms:7.3f} ms per round of {K} step(s)  ({K * a.batch / ms * 1e3:7.0f} utt/s)  delta vs full {ms - ref[K]:+.3f} ms", flush=True)


if __name__ == "__main__":
    main()
