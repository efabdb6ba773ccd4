"""One NT GEMM shape, warm (same operands back to back) and cold (a 512 MB write between launches), event-timed:
python tools/gemm_probe.py M N K [M N K ...]"""
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


def run(M, N, K, reps=30, pad=0):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K + pad, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K + pad, device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda")
    big = torch.empty(128 << 20, device="cuda")
    call = lambda: _cabi.check(L.masr_test_gemm_epi(P(A), K + pad, P(B), K + pad, M, N, K, P(bias), 0, 0.0, P(res), None, P(out), None, S()))
    for _ in range(3):
        call()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); call(); e1.record()
    torch.cuda.synchronize()
    warm = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)[reps // 2]
    for e0, e1 in ev:
        big.fill_(1.0)
        e0.record(); call(); e1.record()
    torch.cuda.synchronize()
    cold = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)[reps // 2]
    print(f"M {M} N {N} K {K} pad {pad}: warm {warm:6.1f} us  cold {cold:6.1f} us   ({2.0 * M * N * K / 1e9:.2f} GFLOP)")


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]] or [640, 512, 2048, 640, 512, 512, 4000, 512, 2048, 4000, 512, 512, 4000, 2048, 512, 640, 2048, 512]
    for i in range(0, len(a), 3):
        for pad in (0, 64, 8):
            run(*a[i:i + 3], pad=pad)
