"""RM (weight-gradient form) GEMM microbenchmark for profiling: dW[M][N] = A[K][M]^T B[K][N]."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(2048, 2048, 1024), (2048, 2048, 4000), (4096, 4096, 4096)]:
    A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
    Cc = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        L.masr_test_gemm(P(A), M, P(B), N, M, N, K, 1, None, 0, P(Cc), N, S())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.masr_test_gemm(P(A), M, P(B), N, M, N, K, 1, None, 0, P(Cc), N, S())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"RM M={M} N={N} K={K}: {us:.1f} us {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s")
