"""What the vendor library (torch.nn.functional.conv2d -> MIOpen, bf16, channels_last) takes for the four 3x3 convolutions of the hkust
front-end, forward / input gradient / weight gradient, graph-replayed: a calibration beside our own conv launches (bench.py roofline.launches).
GPU box only."""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
B, T, D = 16, 1000, 80
shapes = [("conv2 64->64", 64, 64, T, D), ("conv3 64->128", 64, 128, T // 2, D // 2), ("conv4 128->128", 128, 128, T // 2, D // 2)]
def timed(fn, per=20, reps=5):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            for _ in range(per): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * per) * 1e3
    except Exception as ex:                                      # (capture not supported for this algorithm: plain launches, >= 60 us each)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / per * 1e3
for name, ci, co, H, W in shapes:
    x = torch.randn(B, ci, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, co, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    gf = 2.0 * B * H * W * ci * co * 9 / 1e9
    fwd = timed(lambda: F.conv2d(x, w, padding=1))
    dgr = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
    wgr = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
    print(f"{name:16s} {gf:6.1f} GFLOP: forward {fwd:7.1f} us ({gf / fwd * 1e3:6.0f} TFLOP/s)  dgrad {dgr:7.1f} us  wgrad {wgr:7.1f} us")
