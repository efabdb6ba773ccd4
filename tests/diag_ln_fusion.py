"""diagnostic (not collected): which gradient tensors differ between the fused-LayerNorm step and the standalone one"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import masr_amd  # noqa
from masr_amd.engine import MasrEngine
from oracle import ref_cpu
from oracle.make_goldens import TINY, ODIM, synth_batch

cfg = dict(TINY); cfg["dropout"] = cfg["pos_dropout"] = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=5)
xs, il, ys, ol = synth_batch(21, [64, 52, 40, 33], [9, 7, 5, 3])
outs = []
for fused in (True, False):
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.2); eng.load_state_dict(sd); eng.set_seed(99); eng.set_ln_fusion(fused)
    eng.run_batch(xs, il, ys, ol.clone(), train=True)
    torch.cuda.synchronize()
    outs.append((eng.grads.clone(), eng))
(ga, eng), (gb, _) = outs
for n, (off, shape) in eng.table.items():
    k = int(np.prod(shape)); a, b = ga[off:off + k], gb[off:off + k]
    if not torch.equal(a, b):
        d = (a - b).abs()
        print(f"{n:50s} max|diff| {float(d.max()):.3e} rel {float(d.max() / (b.abs().max() + 1e-30)):.2e}  n_diff {int((a != b).sum())}/{k}")
print("done")
