"""GPU diagnostic: per-tensor gradient error of the HIP engine vs the CPU oracle (tiny config)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd
from masr_amd.engine import MasrEngine
from oracle import ref_cpu
from oracle.make_goldens import TINY, ODIM, synth_batch

case = sys.argv[1] if len(sys.argv) > 1 else "ragged"
CASES = {"ragged": ([64, 52, 40, 33], [9, 7, 5, 3]), "same": ([48, 48, 48], [6, 6, 4]), "single": ([37], [5]),
         "ragged_same_olen": ([64, 52, 40, 33], [7, 7, 7, 7]), "same_ilen_ragged_olen": ([64, 64, 64, 64], [9, 7, 5, 3])}
sd = ref_cpu.deterministic_state_dict(TINY, ODIM, seed=7)
xs, il, ys, ol = synth_batch(11, *CASES[case])
eng = MasrEngine(TINY, ODIM, label_smoothing=0.2)
eng.load_state_dict(sd)
eng.run_batch(xs, il, ys, ol, train=True)
print(eng.read_stats())
with ref_cpu.bf16_emulation():
    p = ref_cpu.leafify(sd, TINY)
    info, grads, logit, gold = ref_cpu.run_batch_train(p, TINY, (xs, il, ys, ol.clone()), 0.2)
print(info)
g_all = eng.state_dict(flat=eng.grads)
for n in ref_cpu.grad_param_names(p, TINY):
    a, b = g_all[n].cpu().double(), grads[n].double()
    rel = float((a - b).norm() / (b.norm() + 1e-30))
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print(f"{n:55s} rel {rel:.4f} cos {cos:.5f} |ref| {float(b.norm()):.3e} |got| {float(a.norm()):.3e}")
