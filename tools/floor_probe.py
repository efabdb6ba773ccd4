#!/usr/bin/env python3
"""Per-launch floor of dependent launches on one stream (events around N back-to-back launches)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import masr_amd  # noqa
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
dev = torch.device("cuda:0")
def timeit(tag, fn, n=400):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20): fn(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(2e7))
        e0.record()
        for _ in range(n): fn(s)
        e1.record(); torch.cuda.synchronize()
    print(f"{tag:40s} {e0.elapsed_time(e1) / n * 1e3:7.2f} us / launch")
x1 = torch.zeros(1, device=dev); xk = torch.zeros(592 * 512, device=dev); xm = torch.zeros(4000 * 512, device=dev)
timeit("torch add_ 1 element", lambda s: x1.add_(1.0))
timeit("torch add_ 592x512 fp32", lambda s: xk.add_(1.0))
timeit("torch add_ 4000x512 fp32", lambda s: xm.add_(1.0))
def gemm(M, N, K):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); Cc = torch.zeros(M, N, device=dev)
    return lambda s: _cabi.check(L.masr_test_gemm(P(A), K, P(B), K, M, N, K, 0, None, 0, P(Cc), N, C.c_void_p(s.cuda_stream)), "g")
for M, N, K in ((592, 512, 512), (592, 512, 2048), (592, 2048, 512), (592, 1536, 512), (592, 512, 1536), (64, 64, 512), (64, 64, 2048), (64, 64, 8192),
                (4000, 512, 512), (4000, 512, 2048), (4000, 2048, 512)):
    timeit(f"gemm M={M} N={N} K={K}", gemm(M, N, K))
