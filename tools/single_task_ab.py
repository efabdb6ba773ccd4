#!/usr/bin/env python3
"""One task per GPU: inner-step throughput under the four combinations {k-split off / on} x {kernel-by-kernel launches / step graphs}.
If the k-split's ~70 us of saved kernel time per step do not show in wall time with direct launches but do under graph replay, the
decoder chain is host-enqueue-bound there, not GPU-bound.    python tools/single_task_ab.py [seconds per leg = 1.0]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import masr_amd  # noqa: F401
from masr_amd.engine import MasrEngine
from masr_amd.model import reference_init_state_dict

HK = dict(idim=80, nheads=8, d_model=512, d_inner=2048, dropout=0.1, pos_dropout=0.1, tgt_share_weight=1, encoder=dict(nlayers=2),
          decoder=dict(nlayers=4), meta={"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}})
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
torch.manual_seed(531)
sd = reference_init_state_dict(HK, 367)
g = torch.Generator().manual_seed(0)
B, T = 16, 1000
xs = torch.randn(B, T, 80, generator=g).cuda()
il = torch.full((B,), T, dtype=torch.int64)
ol = torch.randint(10, 41, (B,), generator=g)
ys = [torch.randint(1, 366, (int(n),), generator=g) for n in ol]
eng = MasrEngine(HK, 367, label_smoothing=0.2)
eng.load_state_dict(sd)
mom = torch.zeros_like(eng.params)
lr = 512 ** -0.5 * 25000 ** -0.5


def run(n):
    for i in range(n):
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        eng.clip_sgd_step(mom, 5.0, lr, 0.9, True, False)
    torch.cuda.synchronize()


res = {}
for rep in range(2):
    for graphs in (False, True):
        for ks in (False, True):
            eng.set_ksplit(ks)
            eng.set_step_graphs(graphs)
            run(30)
            t0 = time.perf_counter(); run(50); dt = (time.perf_counter() - t0) / 50
            n = max(50, int(secs / dt))
            t0 = time.perf_counter(); run(n); dt = (time.perf_counter() - t0) / n
            res.setdefault((graphs, ks), []).append(B / dt)
for (graphs, ks), v in res.items():
    print(f"step graphs {'on ' if graphs else 'off'}  k-split {'on ' if ks else 'off'}: " + " / ".join(f"{x:7.0f}" for x in v) + " utt/s")
