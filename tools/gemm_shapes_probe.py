"""Encoder-row NT GEMM shapes of the hkust step through masr_test_gemm (plain fp32 epilogue), operands rotated over 8 sets; beside
tools/gemm_lib_probe.py (the vendor library on the same shapes).  GPU box only."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(M, N, K, iters=40, rot=8):
    sets = [(torch.randn(M, K, device="cuda").bfloat16(), torch.randn(N, K, device="cuda").bfloat16(), torch.zeros(M, N, device="cuda")) for _ in range(rot)]
    call = lambda d: _cabi.check(L.masr_test_gemm(P(d[0]), K, P(d[1]), K, M, N, K, 0, None, 0, P(d[2]), N, S()), "g")
    with torch.cuda.stream(torch.cuda.Stream()):
        for i in range(rot): call(sets[i])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): call(sets[i % rot])
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    ref = sets[0][0].float() @ sets[0][1].float().t()
    err = float((sets[0][2] - ref).abs().max() / ref.abs().max())
    return us, err
out = []
for tag, M, N, K in (("vgg2enc", 4000, 512, 2688), ("qkv", 4000, 1536, 512), ("out", 4000, 512, 512), ("ffn1", 4000, 2048, 512), ("ffn2", 4000, 512, 2048),
                     ("kv_mem", 4000, 4096, 512), ("dqkv", 4000, 512, 1536), ("dkv", 4000, 512, 4096), ("dvgg", 4000, 2688, 512)):
    us, err = run(M, N, K)
    out.append(f"{tag} {us:.1f}" + ("" if err < 1e-3 else f" ERR {err:.1e}"))
print("GEMM", " | ".join(out))
