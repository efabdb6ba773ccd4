"""conv3x3 weight gradient against the map width (tile = 16 x 8 pixels, or 8 x 16 where the width pads better to eights: csrc/conv.hip
wgrad2_tw).  A width of 96 / 48 runs the 16-wide tiles that 83 / 40 / 41 used to round up to, so one run holds both sides of the A/B.
usage: wgrad_widths.py [launches=50]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import masr_amd  # noqa
from masr_amd import _cabi

L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 16
for (H, CI, CO, widths) in ((1000, 64, 64, (80, 83, 88, 96)), (500, 64, 128, (40, 41, 48)), (500, 128, 128, (40, 41, 48))):
    for W in widths:
        x = torch.randn(B, H, W, CI, device="cuda").bfloat16(); dy = torch.randn(B, H, W, CO, device="cuda").bfloat16()
        n = int(L.masr_test_conv3x3_wgrad_slab_floats(B, H, W, CI, CO)); slab = torch.zeros(n, device="cuda"); dw = torch.zeros(CO, CI, 3, 3, device="cuda")
        fn = lambda: L.masr_test_conv3x3_wgrad(P(x), P(dy), P(dw), P(slab), n, B, H, W, CI, CO, S())
        for _ in range(5):
            _cabi.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / N * 1e3
        tw = 8 if ((W + 7) // 8) * 8 < ((W + 15) // 16) * 16 else 16
        th = 128 // tw
        padded = ((W + tw - 1) // tw) * tw * ((H + th - 1) // th) * th
        print(f"wgrad {CI:3d}->{CO:3d} {H} x {W:2d}: {us:7.1f} us (incl. the slab reduce)  tiles {tw:2d} wide: {padded / (H * W):.3f} x the map's pixels, "
              f"{2.0 * 9 * CI * CO * B * H * W / us / 1e6:7.1f} TFLOP/s algorithmic", flush=True)
