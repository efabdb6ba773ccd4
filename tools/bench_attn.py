#!/usr/bin/env python3
"""Attention kernels alone (masr_test_attention = forward + backward of one attention) on the path's three shapes; us per call pair."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import masr_amd  # noqa
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def run(tag, B, H, Tq, Tk, hd, causal, masked, n=200):
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda T: torch.randn(B, T, H, hd, device="cuda", generator=g).bfloat16()
    q, k, v, do = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
    o, dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    lse = torch.zeros(B, H, Tq, device="cuda"); delta = torch.zeros(B, H, Tq, device="cuda")
    klens = torch.full((B,), Tk, device="cuda", dtype=torch.int32) if masked else None
    s = torch.cuda.Stream()
    call = lambda: _cabi.check(L.masr_test_attention(P(q), P(k), P(v), P(do), P(o), P(dq), P(dk), P(dv), P(lse), P(delta), P(klens), B, H, Tq, Tk, hd, causal,
                                                    C.c_void_p(s.cuda_stream)))
    with torch.cuda.stream(s):
        for _ in range(10): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): call()
        e1.record(); torch.cuda.synchronize()
    print(f"{tag:14s} B={B} H={H} Tq={Tq} Tk={Tk}: {e0.elapsed_time(e1) / n * 1e3:7.2f} us fwd + bwd")
run("encoder self", 16, 8, 250, 250, 64, 0, True)
run("decoder cross", 16, 8, 37, 250, 64, 0, True)
run("decoder self", 16, 8, 37, 37, 64, 1, False)
def run_fwd(B, H, Tq, Tk, hd=64, n=300):
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda T: torch.randn(B, T, H, hd, device="cuda", generator=g).bfloat16()
    q, k, v = mk(Tq), mk(Tk), mk(Tk)
    o = torch.zeros_like(q); lse = torch.zeros(B, H, Tq, device="cuda")
    s = torch.cuda.Stream()
    call = lambda: _cabi.check(L.masr_test_attention_dropout(P(q), P(k), P(v), P(o), P(lse), B, H, Tq, Tk, hd, C.c_float(0.0), 1, 1, C.c_void_p(s.cuda_stream)))
    with torch.cuda.stream(s):
        for _ in range(10): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): call()
        e1.record(); torch.cuda.synchronize()
    print(f"fwd only B={B} H={H} Tq={Tq} Tk={Tk}: {e0.elapsed_time(e1) / n * 1e3:7.2f} us")
for Tk in (64, 128, 256, 512, 1024): run_fwd(16, 8, 250, Tk)
for Tq in (64, 128, 512, 1024): run_fwd(16, 8, Tq, 256)
run_fwd(1, 1, 64, 64); run_fwd(1, 1, 64, 1024); run_fwd(2, 8, 250, 250)
