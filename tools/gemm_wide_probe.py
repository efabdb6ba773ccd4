"""The wide bf16-output NT GEMM launches of the encoder rows (K = 512) through masr_test_gemm_epi: time per launch and a checksum of the
output (run once per setting of the kernel selection and compare).  GPU box only."""
import ctypes as C, hashlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import masr_amd
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(tag, M, N, K, flavour, iters=40, rot=6):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    sets = []
    for _ in range(rot):
        A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16(); B = (torch.randn(N, K, device="cuda", generator=g) * 0.5).bfloat16()
        bias = torch.randn(N, device="cuda", generator=g) if flavour.startswith("bias") else None
        mask = (torch.randn(M, N, device="cuda", generator=g) > 0).bfloat16() if flavour == "mask" else None
        sets.append((A, B, bias, mask, torch.zeros(M + 1, N, device="cuda").bfloat16()))
    drop = 0.1 if flavour == "bias_relu" else 0.0
    call = lambda d: _cabi.check(L.masr_test_gemm_epi(P(d[0]), K, P(d[1]), K, M, N, K, P(d[2]), 1 if flavour == "bias_relu" else 0, drop, None, P(d[3]), None, P(d[4]), S()), "g")
    with torch.cuda.stream(torch.cuda.Stream()):
        for d in sets: call(d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): call(sets[i % rot])
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    d = sets[0]
    ref = d[0].float() @ d[1].float().t()
    if d[2] is not None: ref = ref + d[2]
    if flavour == "bias_relu": ref = ref.clamp_min(0)
    if d[3] is not None: ref = ref * d[3].float()
    out = d[4][:M].float()
    if drop:
        keep = out != 0
        err = float(((out * (1 - drop) - ref) * keep).abs().max() / ref.abs().max())
    else:
        err = float((out - ref).abs().max() / ref.abs().max())
    h = hashlib.md5(d[4].view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:8]
    return f"{tag} {us:.1f}us {h}" + ("" if err < 2e-2 else f" ERR {err:.1e}")
print(" | ".join(run(*a) for a in (("qkv", 4000, 1536, 512, "bias"), ("ffn1", 4000, 2048, 512, "bias_relu"), ("kv", 4000, 4096, 512, "bias"),
                                    ("dffn2", 4000, 2048, 512, "mask"), ("dvgg", 4000, 2688, 512, "plain"), ("small", 1000, 1024, 256, "bias"))))
