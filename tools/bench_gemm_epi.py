"""What each fused epilogue stage of the NT GEMM costs per launch, on the encoder-row and decoder-row shapes of the hkust step."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(tag, M, N, K, bias=False, relu=0, drop=0.0, res=False, mask=False, c32=True, c16=False, iters=30, rotate=1):
    """rotate > 1: cycle through that many operand sets (defeats L2 residency of the activations, as in the real step)"""
    sets = []
    for _ in range(rotate):
        sets.append(dict(A=torch.randn(M, K, device="cuda").bfloat16(), B=torch.randn(N, K, device="cuda").bfloat16(),
                         bias=torch.randn(N, device="cuda") if bias else None, res=torch.randn(M, N, device="cuda") if res else None,
                         mask=torch.randn(M, N, device="cuda").bfloat16() if mask else None,
                         C32=torch.zeros(M, N, device="cuda") if c32 else None, C16=torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if c16 else None))
    def call(d):
        _cabi.check(L.masr_test_gemm_epi(P(d["A"]), K, P(d["B"]), K, M, N, K, P(d["bias"]), relu, drop, P(d["res"]), P(d["mask"]), P(d["C32"]), P(d["C16"]), S()), "gemm")
    with torch.cuda.stream(torch.cuda.Stream()):
        for i in range(3): call(sets[i % rotate])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): call(sets[i % rotate])
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{tag:34s} M={M:5d} N={N:5d} K={K:5d} {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")


for M in (4000, 496):
    for rot in (1, 8):
        print(f"--- M = {M}, operand sets = {rot}")
        run("plain fp32 out", M, 512, 512, rotate=rot)
        run("+bias", M, 512, 512, bias=True, rotate=rot)
        run("+bias +residual", M, 512, 512, bias=True, res=True, rotate=rot)
        run("+bias +dropout", M, 512, 512, bias=True, drop=0.1, rotate=rot)
        run("+bias +dropout +residual (out-proj)", M, 512, 512, bias=True, drop=0.1, res=True, rotate=rot)
        run("bf16 out +bias (qkv)", M, 1536, 512, bias=True, c32=False, c16=True, rotate=rot)
        run("bf16 out +bias +relu +drop (ffn1)", M, 2048, 512, bias=True, relu=1, drop=0.1, c32=False, c16=True, rotate=rot)
        run("ffn2: K=2048 +bias+drop+res", M, 512, 2048, bias=True, drop=0.1, res=True, rotate=rot)
        run("dgrad +mask bf16 out", M, 2048, 512, mask=True, c32=False, c16=True, rotate=rot)
