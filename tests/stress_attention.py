"""Diagnostic (not collected by pytest): repeat the attention / LayerNorm kernels on random shapes and check that every repeat is
bit-identical (a race shows up as run-to-run differences) and close to the torch reference."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import masr_amd  # noqa
from masr_amd import _cabi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_hip_kernels import _attn_ref, P, S
from diag_attn_accuracy import ref64
L = _cabi.lib()
rng = np.random.default_rng(0)
bad = 0
for it in range(150):
    hd = int(rng.choice([16, 32, 64])); H = int(rng.choice([2, 4, 8])); B = int(rng.integers(1, 5))
    Tq = int(rng.integers(1, 300)); causal = int(rng.integers(0, 2)); Tk = Tq if causal else int(rng.integers(1, 400)); masked = (not causal) and bool(rng.integers(0, 2))
    g = torch.Generator(device="cuda").manual_seed(it)
    mk = lambda T: torch.randn(B, T, H, hd, device="cuda", generator=g).bfloat16()
    q, k, v, do = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
    klens = torch.randint(1, Tk + 1, (B,), device="cuda", generator=g).int() if masked else None
    outs = []
    for rep in range(6):
        o, dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        lse = torch.zeros(B, H, Tq, device="cuda"); delta = torch.zeros(B, H, Tq, device="cuda")
        _cabi.check(L.masr_test_attention(P(q), P(k), P(v), P(do), P(o), P(dq), P(dk), P(dv), P(lse), P(delta), P(klens) if masked else None, B, H, Tq, Tk, hd, causal, S()))
        torch.cuda.synchronize()
        outs.append((o, dq, dk, dv, lse))
    same = all(all(torch.equal(a, b) for a, b in zip(outs[0], oo)) for oo in outs[1:])
    kl = klens.cpu() if masked else torch.full((B,), Tk)
    refs = ref64(q, k, v, do, kl, causal)
    err = [float((a.double() - r).pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-30)) for a, r in zip(outs[0][:4], refs)]
    ok = same and max(err) < 0.01 and all(torch.isfinite(a.float()).all() for a in outs[0])
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} hd={hd} H={H} B={B} Tq={Tq} Tk={Tk} causal={causal} masked={masked} repeat-identical={same} rel rms err o/dq/dk/dv {err[0]:.4f} {err[1]:.4f} {err[2]:.4f} {err[3]:.4f}")
print("bad:", bad)
