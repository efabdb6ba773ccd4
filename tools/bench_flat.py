#!/usr/bin/env python3
"""Flat fp32 passes at the path's parameter count (24.9 M): masr_sgd_step against torch's own elementwise kernels, TB/s of p + g read, p written."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import masr_amd  # noqa
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
n = 24_881_455
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda")
def timeit(tag, fn, nbytes, it=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    print(f"{tag:34s} {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s")
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
timeit("masr_sgd_step (p -= lr g)", lambda: _cabi.check(L.masr_sgd_step(P(p), P(g), None, n, C.c_float(1e-6), C.c_float(0.0), 0, 0, S())), 12 * n)
timeit("torch p.add_(g, alpha)", lambda: p.add_(g, alpha=-1e-6), 12 * n)
timeit("torch p.mul_(1.0)", lambda: p.mul_(1.0), 8 * n)
timeit("torch g.square().sum()", lambda: torch.linalg.vector_norm(g), 4 * n)
timeit("torch p.to(bf16)", lambda: p.bfloat16(), 6 * n)
timeit("torch copy", lambda: p.copy_(g), 8 * n)
# cold: rotate over 6 buffer pairs (1.2 GB > the 256 MB infinity cache)
ps = [torch.randn(n, device="cuda") for _ in range(6)]; gs = [torch.randn(n, device="cuda") for _ in range(6)]
cnt = [0]
def cold_sgd():
    i = cnt[0] % 6; cnt[0] += 1
    _cabi.check(L.masr_sgd_step(P(ps[i]), P(gs[i]), None, n, C.c_float(1e-6), C.c_float(0.0), 0, 0, S()))
def cold_torch():
    i = cnt[0] % 6; cnt[0] += 1
    ps[i].add_(gs[i], alpha=-1e-6)
def cold_norm():
    i = cnt[0] % 6; cnt[0] += 1
    torch.linalg.vector_norm(gs[i])
def cold_cast():
    i = cnt[0] % 6; cnt[0] += 1
    ps[i].bfloat16()
timeit("cold masr_sgd_step", cold_sgd, 12 * n, it=60)
timeit("cold torch add_", cold_torch, 12 * n, it=60)
timeit("cold torch norm", cold_norm, 4 * n, it=60)
timeit("cold torch cast bf16", cold_cast, 6 * n, it=60)
