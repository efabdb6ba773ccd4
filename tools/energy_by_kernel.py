#!/usr/bin/env python3
"""Energy per launch of the step's kernel classes (the four-slot headline sits at the board power cap, so a kernel's cost is
its joules).  Each kernel is launched back to back for a few seconds at the headline shapes (B = 16, T = 1000, idim 80) while the board
power is sampled from sysfs (bench.py's ClockSampler); energy per launch = (power - idle) x time per launch, set beside the MFMAs' own
energy (algorithmic FLOPs x 0.58 pJ, tools/energy/energy_probe.hip) and the HBM-byte energy at 5 pJ / bit.  GPU box only.

    python tools/energy_by_kernel.py [seconds per kernel = 2.5]"""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import masr_amd  # noqa: F401
from masr_amd import _cabi
from bench import ClockSampler

L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
PJ_FLOP, PJ_BIT = 0.58, 5.0


def measure(name, fn, flops, hbm_bytes):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    n = max(50, int(secs / (us * 1e-6)))
    with ClockSampler(0, period=0.1) as cs:
        t0 = time.perf_counter()
        done = 0
        while done < n:
            for _ in range(min(200, n - done)):
                fn()
            done += 200
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    pw = [x[2] for x in cs.samples if x[2] is not None]
    pw = pw[len(pw) // 4:]                                   # (drop the ramp)
    p = float(np.median(pw)) if pw else float("nan")
    us = dt / done * 1e6
    e = (p - IDLE) * us * 1e-6                               # joules per launch above idle
    mf, hb = flops * PJ_FLOP * 1e-12, hbm_bytes * 8 * PJ_BIT * 1e-12
    print(f"{name:44s} {us:8.1f} us  {p:6.0f} W  {e * 1e3:7.2f} mJ/launch   MFMA-own {mf * 1e3:6.2f} mJ  HBM {hb * 1e3:5.2f} mJ  other {(e - mf - hb) * 1e3:6.2f} mJ", flush=True)
    return e


with ClockSampler(0, period=0.1) as cs0:
    time.sleep(1.5)
IDLE = float(np.median([x[2] for x in cs0.samples if x[2] is not None]))
print(f"idle {IDLE:.0f} W; energy above idle per launch, B = 16 x T = 1000 x idim 80 (one inner step launches each of these once unless noted)")
tot = 0.0
B = 16
# ---- convs
for name, (H, W, CI, CO), pool in (("conv2 fwd 64->64 + pool", (1000, 80, 64, 64), True), ("conv3 fwd 64->128", (500, 40, 64, 128), False),
                                   ("conv4 fwd 128->128 + pool", (500, 40, 128, 128), True), ("conv3 dgrad 128->64", (500, 40, 128, 64), False)):
    x = rnd(B, H, W, CI).bfloat16(); wk = (rnd(CO, 9 * CI) * 0.05).bfloat16(); bias = rnd(CO)
    out = torch.zeros(B, H, W, CO, device="cuda").bfloat16()
    fl = 2.0 * 9 * CI * CO * B * H * W
    if pool:
        po = torch.zeros(B, H // 2, W // 2, CO, device="cuda").bfloat16(); idx = torch.zeros(B, H // 2, W // 2, CO, device="cuda", dtype=torch.uint8)
        fn = lambda x=x, wk=wk, bias=bias, out=out, po=po, idx=idx, H=H, W=W, CI=CI, CO=CO: L.masr_test_conv3x3_pool_idx(P(x), P(wk), P(bias), P(out), P(po), P(idx), 1, B, H, W, CI, CO, S())
        by = 2.0 * B * H * W * CI + 3.0 * B * (H // 2) * (W // 2) * CO
    else:
        fn = lambda x=x, wk=wk, bias=bias, out=out, H=H, W=W, CI=CI, CO=CO: L.masr_test_conv3x3(P(x), P(wk), P(bias), 1, P(out), B, H, W, CI, CO, S())
        by = 2.0 * B * H * W * (CI + CO)
    tot += measure(name, fn, fl, by)
for name, (H, W, CI, CO) in (("conv2 wgrad 64x64", (1000, 80, 64, 64)), ("conv4 wgrad 128x128", (500, 40, 128, 128))):
    x = rnd(B, H, W, CI).bfloat16(); dy = rnd(B, H, W, CO).bfloat16()
    n = int(L.masr_test_conv3x3_wgrad_slab_floats(B, H, W, CI, CO)); slab = torch.zeros(n, device="cuda"); dw = torch.zeros(CO, CI, 3, 3, device="cuda")
    fn = lambda x=x, dy=dy, dw=dw, slab=slab, n=n, H=H, W=W, CI=CI, CO=CO: L.masr_test_conv3x3_wgrad(P(x), P(dy), P(dw), P(slab), n, B, H, W, CI, CO, S())
    tot += measure(name + " (full-resolution dy)", fn, 2.0 * 9 * CI * CO * B * H * W, 2.0 * B * H * W * (CI + CO))
# ---- conv1 forward
x1 = rnd(B, 1000, 80); w1 = rnd(64, 9) * 0.3; b1 = rnd(64); o1 = torch.zeros(B, 1000, 80, 64, device="cuda").bfloat16(); bits = torch.zeros(B, 1000, 80, device="cuda", dtype=torch.int64)
tot += measure("conv1 fwd (fp32 MFMA)", lambda: L.masr_test_conv1_fwd(P(x1), P(w1), P(b1), P(o1), P(bits), B, 1000, 80, S()), 0.0, B * 1000 * 80 * (4 + 128 + 8))
# ---- GEMMs over the 4000 encoder rows
for name, (M, N, K), times in (("GEMM FFN1 4000x2048x512 (fp32 out)", (4000, 2048, 512), "x4 per step"), ("GEMM q/k/v 4000x1536x512", (4000, 1536, 512), "x2"),
                               ("GEMM vgg2enc 4000x512x2560", (4000, 512, 2560), "x2"), ("GEMM K|V 4000x4096x512", (4000, 4096, 512), "x1")):
    A = rnd(M, K).bfloat16(); Bm = rnd(N, K).bfloat16(); Cm = torch.zeros(M, N, device="cuda")
    fn = lambda A=A, Bm=Bm, Cm=Cm, M=M, N=N, K=K: L.masr_test_gemm(P(A), K, P(Bm), K, M, N, K, 0, None, 0, P(Cm), N, S())
    measure(f"{name} [{times}]", fn, 2.0 * M * N * K, 2.0 * (M * K + N * K) + 4.0 * M * N)
# ---- grouped weight gradient (one member pair of the step's 38)
rows, N, K = 4000, 2048, 512
dy = rnd(rows, N).bfloat16(); xx = rnd(rows, K).bfloat16(); dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
measure("wgrad grouped: dW 2048x512 over 4000 rows", lambda: L.masr_test_wgrad_grouped(P(dy), N, P(xx), K, P(dW), P(db), None, None, rows, N, K, S()),
        2.0 * rows * N * K, 2.0 * rows * (N + K) + 4.0 * N * K)
# ---- attention (encoder shape) forward + backward
Bq, H, T, hd = 16, 8, 250, 64
mk = lambda: rnd(Bq, T, H, hd).bfloat16()
q, k, v, do = mk(), mk(), mk(), mk()
o, dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
lse = torch.zeros(Bq, H, T, device="cuda"); delta = torch.zeros(Bq, H, T, device="cuda")
measure("attention enc fwd + bwd (16 x 8 x 250^2) [x2]", lambda: L.masr_test_attention(P(q), P(k), P(v), P(do), P(o), P(dq), P(dk), P(dv), P(lse), P(delta), None, Bq, H, T, T, hd, 0, S()),
        3 * 4.0 * Bq * H * T * T * hd, 8 * 2.0 * Bq * T * H * hd)
# ---- LayerNorm forward + backward, encoder rows
rows, E = 4000, 512
x = rnd(rows, E); gm = rnd(E); bt = rnd(E); dyl = rnd(rows, E)
y = torch.zeros(rows, E, device="cuda"); y16 = torch.zeros(rows, E, device="cuda").bfloat16(); mean = torch.zeros(rows, device="cuda"); rstd = torch.zeros(rows, device="cuda")
dx = torch.zeros(rows, E, device="cuda"); dx16 = torch.zeros(rows, E, device="cuda").bfloat16(); dgm = torch.zeros(E, device="cuda"); dbt = torch.zeros(E, device="cuda")
slab = torch.zeros(int(L.masr_test_layernorm_slab_floats(rows, E)), device="cuda")
measure("LayerNorm fwd + bwd 4000 x 512 [x5]", lambda: L.masr_test_layernorm(P(x), P(gm), P(bt), P(dyl), P(y), P(y16), P(mean), P(rstd), P(dx), P(dx16), P(dgm), P(dbt), P(slab), rows, E,
                                                                            C.c_float(0.0), 0, 0, S()), 0.0, rows * E * (4 + 4 + 2 + 4 + 4 + 4 + 2))
# ---- flat optimiser pass
n = 24_881_455
p = rnd(n); gg = rnd(n); mo = torch.zeros(n, device="cuda")
measure("SGD pass with momentum (24.9 M params)", lambda: L.masr_sgd_step(P(p), P(gg), P(mo), n, C.c_float(1e-6), C.c_float(0.9), 1, 0, S()), 0.0, 20.0 * n)
print(f"(sum of the nine conv-class rows above: {tot * 1e3:.1f} mJ; a 16-utterance step costs 2.2 J at the board, 1.7-1.85 J above idle)")
