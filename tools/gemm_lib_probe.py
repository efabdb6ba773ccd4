"""What the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) takes for the encoder-row GEMM shapes of the hkust step:
a calibration of what these shapes can reach on the chip, beside our own kernels' times (tools/step_timeline.py).  GPU box only."""
import sys, torch
import torch.nn.functional as F
shapes = [("vgg2enc", 4000, 512, 2688), ("qkv", 4000, 1536, 512), ("out_proj", 4000, 512, 512), ("ffn1", 4000, 2048, 512), ("ffn2", 4000, 512, 2048),
          ("kv_mem", 4000, 4096, 512), ("dec_ffn2", 592, 512, 2048), ("dec_qkv", 592, 1536, 512)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    # (a Python loop of launches is host-bound at ~18 us per call: the calls are captured into one graph and replayed)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5): F.linear(a, w)
    torch.cuda.synchronize()
    graph, per = torch.cuda.CUDAGraph(), 50
    with torch.cuda.graph(graph, stream=side):
        for _ in range(per): out = F.linear(a, w)
    graph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): graph.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (reps * per) * 1e3
    print(f"{name:10s} M {M} N {N} K {K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
