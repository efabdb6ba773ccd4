"""Diagnostic (not collected): error of the attention kernels against a float64 reference on the same bf16 operands (mean / max abs error
of o, dq, dk, dv relative to the reference's rms), for the path's head dims and a few score scales."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import masr_amd  # noqa
from masr_amd import _cabi
from test_hip_kernels import S
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
def ref64(q, k, v, do, klens, causal):
    q, k, v, do = [t.double() for t in (q, k, v, do)]
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    B, Tq, H, hd = q.shape; Tk = k.shape[1]
    s = torch.einsum('bqhd,bkhd->bhqk', q, k) / hd ** 0.5
    mask = torch.zeros(B, 1, Tq, Tk, dtype=torch.bool, device=q.device)
    for b in range(B): mask[b, :, :, int(klens[b]):] = True
    if causal: mask |= torch.triu(torch.ones(Tq, Tk, dtype=torch.bool, device=q.device), 1)
    p = torch.softmax(s.masked_fill(mask, -1e300), -1)
    o = torch.einsum('bhqk,bkhd->bqhd', p, v)
    o.backward(do)
    return o.detach(), q.grad, k.grad, v.grad
for hd, H, B, Tq, Tk, causal in () if __name__ != '__main__' else ((16, 4, 4, 8, 8, 0), (16, 4, 4, 9, 9, 1), (16, 4, 2, 70, 70, 0), (64, 8, 4, 250, 250, 0), (64, 8, 4, 37, 250, 0), (64, 8, 4, 37, 37, 1)):
    for scale in (1.0, 3.0):
        g = torch.Generator(device="cuda").manual_seed(7)
        mk = lambda T, m=1.0: (torch.randn(B, T, H, hd, device="cuda", generator=g) * m).bfloat16()
        q, k, v, do = mk(Tq, scale), mk(Tk, scale), mk(Tk), mk(Tq)
        klens = torch.full((B,), Tk, device="cuda", dtype=torch.int32)
        o, dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        lse = torch.zeros(B, H, Tq, device="cuda"); delta = torch.zeros(B, H, Tq, device="cuda")
        _cabi.check(L.masr_test_attention(P(q), P(k), P(v), P(do), P(o), P(dq), P(dk), P(dv), P(lse), P(delta), P(klens), B, H, Tq, Tk, hd, causal, S()))
        refs = ref64(q, k, v, do, klens.cpu(), causal)
        out = []
        for a, r in zip((o, dq, dk, dv), refs):
            d = (a.double() - r); out.append(f"{float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()):.4f}")
        print(f"hd={hd} Tq={Tq} Tk={Tk} causal={causal} qk-scale={scale}: rel rms err o/dq/dk/dv " + " ".join(out))
