"""What torch's scaled_dot_product_attention (the vendor flash-attention path, bf16) takes at the hkust attention shapes, forward and
forward + backward, graph-replayed where capture works: a calibration beside attn_fwd_ring / attn_bwd_ring (tools/step_timeline.py).
GPU box only."""
import torch
import torch.nn.functional as F
def timed(fn, per=20, reps=5):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * per) * 1e3, "graph"
    except Exception:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / per * 1e3, "launches"
for name, B, H, Tq, Tk, hd, causal in (("encoder self", 16, 8, 250, 250, 64, False), ("decoder self", 16, 8, 37, 37, 64, True), ("decoder cross", 16, 8, 37, 250, 64, False)):
    for p in (0.0, 0.1):
        q = torch.randn(B, H, Tq, hd, device="cuda").bfloat16().requires_grad_(True)
        k = torch.randn(B, H, Tk, hd, device="cuda").bfloat16().requires_grad_(True)
        v = torch.randn(B, H, Tk, hd, device="cuda").bfloat16().requires_grad_(True)
        do = torch.randn(B, H, Tq, hd, device="cuda").bfloat16()
        def fwd():
            with torch.no_grad():
                return F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=causal)
        def fb():
            o = F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=causal)
            o.backward(do)
            q.grad = k.grad = v.grad = None
        tf, hf = timed(fwd); tb, hb = timed(fb)
        print(f"{name:14s} dropout {p}: forward {tf:6.1f} us ({hf}), forward + backward {tb:6.1f} us ({hb})")
