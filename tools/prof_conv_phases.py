"""Where do a conv workgroup's cycles go?  Runs the patch conv with its phase-timing hooks on (wave 0 of every
workgroup stamps s_memtime at phase boundaries) and prints the mean cycles per phase next to the un-instrumented launch time."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import masr_amd  # noqa
from masr_amd import _cabi

L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
import os
STREAM = not (os.environ.get("MASR_CONV_STREAM") == "0")
NAMES = (["MFMA wave: start-up", "MFMA wave: tap loops", "MFMA wave: epilogues", "weight wave: waiting", "patch wave: issuing", "patch wave: waiting"]
         if STREAM else ["patch load+stage", "weight prefetch issue", "frag reads + MFMA", "weight stage (LDS store)", "barrier", "epilogue"])
shapes = [(16, 1000, 80, 64, 64, 16, 16), (16, 500, 40, 64, 128, 16, 8), (16, 500, 40, 128, 128, 16, 8), (16, 500, 40, 128, 64, 16, 8)]
for (B, H, W, CI, CO, TH, TW) in shapes:
    x = torch.randn(B, H, W, CI, device="cuda").bfloat16()
    wk = (torch.randn(CO, 9 * CI, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(CO, device="cuda")
    out = torch.zeros(B, H, W, CO, device="cuda").bfloat16()
    nwg = ((W + TW - 1) // TW) * ((H + TH - 1) // TH) * B
    prof = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda")

    def timed(fn):
        for _ in range(3):
            _cabi.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    us = timed(lambda: L.masr_test_conv3x3(P(x), P(wk), P(bias), 1, P(out), B, H, W, CI, CO, S()))
    usp = timed(lambda: L.masr_test_conv3x3_prof(P(x), P(wk), P(bias), 1, P(out), B, H, W, CI, CO, P(prof), S()))
    if STREAM:
        abl = {n: timed(lambda f=f: L.masr_test_conv3x3_prof(P(x), P(wk), P(bias), 1 | f << 8, P(out), B, H, W, CI, CO, P(prof), S()))
               for n, f in (("no patch DMA", 1), ("no weight DMA", 2), ("no epilogue", 4), ("MFMA loop only", 7))}
        usp = timed(lambda: L.masr_test_conv3x3_prof(P(x), P(wk), P(bias), 1, P(out), B, H, W, CI, CO, P(prof), S()))
        abl_line = "    ablations (us): " + ", ".join(f"{n} {v:.1f}" for n, v in abl.items())
    else:
        abl_line = None
    pr = prof.view(nwg, 6).double()
    if STREAM:
        pr = pr[pr[:, 1] > 0]                     # persistent grid: only the launched workgroups wrote
        nwg = pr.shape[0]
        mean, tot = pr.mean(0), pr[:, :3].sum(1)  # the MFMA wave's three phases add up to the workgroup's life
    else:
        mean, tot = pr.mean(0), pr.sum(1)
    print(f"conv {CI}->{CO} {H}x{W}: {us:.1f} us plain, {usp:.1f} us instrumented, {nwg} workgroups, "
          f"{tot.mean():.0f} cycles/WG (min {tot.min():.0f} max {tot.max():.0f})")
    for k in range(6):
        print(f"    {NAMES[k]:26s} {mean[k]:9.0f} cycles  {100 * mean[k] / tot.mean():5.1f}%")
    if abl_line:
        print(abl_line)
