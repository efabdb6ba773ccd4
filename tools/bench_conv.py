"""Time the 3x3 conv forward/dgrad kernel at the B=16, T=1000, D=80 shapes (HIP events, 20 launches each).
usage: bench_conv.py [batch=16] [launches=20]   (batch 64 x 400 launches = back-to-back 0.35 ms dispatches for a clock pass)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import masr_amd  # noqa
from masr_amd import _cabi

L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
BB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = [(BB, 1000, 80, 64, 64), (BB, 500, 40, 64, 128), (BB, 500, 40, 128, 128), (BB, 500, 40, 128, 64)]
for (B, H, W, CI, CO) in shapes:
    x = torch.randn(B, H, W, CI, device="cuda").bfloat16()
    wk = (torch.randn(CO, 9 * CI, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(CO, device="cuda")
    out = torch.zeros(B, H, W, CO, device="cuda").bfloat16()
    for _ in range(3):
        _cabi.check(L.masr_test_conv3x3(P(x), P(wk), P(bias), 1, P(out), B, H, W, CI, CO, S()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        L.masr_test_conv3x3(P(x), P(wk), P(bias), 1, P(out), B, H, W, CI, CO, S())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / N
    fl = 2.0 * 9 * CI * CO * B * H * W
    by = 2.0 * B * H * W * (CI + CO)
    print(f"conv {CI:3d}->{CO:3d} {H}x{W}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({by / ms / 1e9:6.2f} TB/s algorithmic)  checksum {float(out.float().sum()):.4e}")
