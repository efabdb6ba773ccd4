"""What bounds the encoder-row weight-gradient launch: the SAME tile work with operands that stay resident (many group members over one
pair of panels) against operands that do not (the engine's own launch).  python tools/wgrad_probe.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd  # noqa
from masr_amd import _cabi

L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(rows, N, K, members, reps=20, shared_out=True):
    g = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(rows, N, device="cuda", generator=g).bfloat16()
    x = torch.randn(rows, K, device="cuda", generator=g).bfloat16()
    dW = torch.zeros((1 if shared_out else members) * N * K, device="cuda")
    stride = 0 if shared_out else N * K
    for _ in range(3):
        _cabi.check(L.masr_test_wgrad_grouped_n(P(dy), N, P(x), K, P(dW), stride, members, 0, rows, rows, N, K, S()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _cabi.check(L.masr_test_wgrad_grouped_n(P(dy), N, P(x), K, P(dW), stride, members, 0, rows, rows, N, K, S()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * rows * N * K * members
    tiles = members * ((N + 255) // 256) * ((K + 255) // 256)
    print(f"rows {rows} N {N} K {K} x {members} members: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  ({tiles} tiles of 256 x 256; operands {(rows * (N + K) * 2) / 1e6:.1f} MB)")


if __name__ == "__main__":
    run(4000, 512, 512, 64)               # 256 tiles over 8 MB of operands: everything resident
    run(4000, 2048, 2048, 4)              # 256 tiles over 32 MB
    run(4000, 2048, 4096, 2)              # 256 tiles over 49 MB
    run(4000, 512, 512, 16)               # 64 tiles: a quarter of the chip
    run(1000, 512, 512, 64)
