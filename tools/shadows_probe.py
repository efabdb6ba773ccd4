"""The operand-shadow tile pass alone (masr_test_linear_shadows: W fp32 [N][K] -> bf16 [N][K] and its transpose): HBM rate per shape and
misalignment of the tensor in the flat buffer.  8 bytes move per element (4 in, 2 + 2 out).  usage: shadows_probe.py [launches=50]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import masr_amd  # noqa
from masr_amd import _cabi

L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for (N, K, src) in ((8192, 2048, 4), (8192, 2048, 5), (2048, 512, 7), (512, 2048, 6), (512, 512, 4), (367, 512, 9), (4096, 514, 4)):
    buf = torch.randn(src + N * K + 64, device="cuda")
    k16 = torch.zeros(N, K, device="cuda").bfloat16(); t16 = torch.zeros(K, N, device="cuda").bfloat16()
    fn = lambda: L.masr_test_linear_shadows(P(buf), src, N, K, N, P(k16), P(t16), S())
    for _ in range(5):
        _cabi.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"W [{N}][{K}] at dword offset {src}: {us:7.1f} us  {8.0 * N * K / us / 1e6:5.2f} TB/s", flush=True)
