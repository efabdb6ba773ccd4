"""GPU micro-benchmark of masr_test_gemm on the shapes of the hkust inner step (B=16, T'=250, L=41)."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(M, N, K, rm, iters=20):
    if rm:
        A = torch.randn(K, (M + 7) // 8 * 8, device="cuda").bfloat16(); B = torch.randn(K, (N + 7) // 8 * 8, device="cuda").bfloat16()
        lda, ldb = A.shape[1], B.shape[1]
    else:
        A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16(); lda = ldb = K
    Cc = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        L.masr_test_gemm(P(A), lda, P(B), ldb, M, N, K, rm, None, 0, P(Cc), N, S())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.masr_test_gemm(P(A), lda, P(B), ldb, M, N, K, rm, None, 0, P(Cc), N, S())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{'RM' if rm else 'NT'} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")

for shp in [(4000,1536,512),(4000,512,512),(4000,2048,512),(4000,512,2048),(4000,512,2560),(4000,2560,512),(4000,1024,512),
            (656,1536,512),(656,512,512),(656,2048,512),(656,512,2048),(656,367,512),(656,512,384),(8192,8192,8192),(4096,4096,4096)]:
    run(*shp, 0)
for shp in [(1536,512,4000),(512,512,4000),(2048,512,4000),(512,2048,4000),(512,2560,4000),(1536,512,656),(512,512,656),(2048,512,656),(512,2048,656),(4096,4096,4096)]:
    run(*shp, 1)
