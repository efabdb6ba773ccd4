#!/usr/bin/env python3
"""How many HIP streams does the chip serve at once?  K streams, each with a chain of N dependent single-workgroup kernels (a 64 x 64
GEMM with a long k loop: tens of microseconds on ONE compute unit, so any number of them fit side by side) queued AHEAD of the GPU;
per stream, the GPU time at which its chain ends.  Streams that are served concurrently finish together; a stream that has to wait
for a place finishes one chain later.

    python tools/queue_probe.py            (GPU_MAX_HW_QUEUES=8 as pretrain.py / bench.py set it; try =4, =16)"""
import ctypes as C
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import masr_amd  # noqa: F401
from masr_amd import _cabi

L = _cabi.lib()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
M = Nn = 64
Kd = int(os.environ.get("PROBE_K", "8192"))
N = int(os.environ.get("PROBE_N", "200"))
A = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(8)]
B = [torch.randn(Nn, Kd, device=dev).bfloat16() for _ in range(8)]
Cc = [torch.zeros(M, Nn, device=dev) for _ in range(8)]


def chain(k, s):
    h = C.c_void_p(s.cuda_stream)
    for _ in range(N):
        _cabi.check(L.masr_test_gemm(P(A[k]), Kd, P(B[k]), Kd, M, Nn, Kd, 0, None, 0, P(Cc[k]), Nn, h))


print(f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}, chains of {N} single-workgroup kernels")
for K in ([int(os.environ['PROBE_STREAMS'])] if os.environ.get('PROBE_STREAMS') else (1, 2, 3, 4, 5, 6, 8)):
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(K - 1)]
    for rep in range(2):
        torch.cuda.synchronize()
        gate = torch.cuda.Event(); start = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        torch.cuda._sleep(int(6e7))                              # ~30 ms: every chain is queued before the gate opens
        start.record(streams[0]); gate.record(streams[0])
        for k, s in enumerate(streams):
            s.wait_event(gate)
            chain(k, s)
            ends[k].record(s)
        torch.cuda.synchronize()
    t = sorted(start.elapsed_time(e) for e in ends)
    print(f"{K} streams: chains end after " + ", ".join(f"{x:6.2f}" for x in t) + " ms")
