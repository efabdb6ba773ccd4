"""The same inner step (hkust geometry: VGG front-end 1->64->64 pool ->128->128 pool, Linear to 512, 2 encoder / 4 decoder Transformer layers,
label-smoothed CE, clip 5, SGD momentum 0.9 nesterov) written with stock torch.nn modules and run by PyTorch-ROCm on the GPU under bf16
autocast: what the vendor libraries (MIOpen, hipBLASLt, the flash-attention path) give for this model on the same chip, beside bench.py's
single_task.  A calibration only -- nothing here is part of the product, and nothing of the product or the oracle is imported.  GPU box only."""
import math, time, torch, torch.nn as nn, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
B, T, D, E, H, FF, NE, ND, C, L = 16, 1000, 80, 512, 8, 2048, 2, 4, 367, 41
class Model(nn.Module):
    def __init__(s):
        super().__init__()
        s.vgg = nn.Sequential(nn.Conv2d(1, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2),
                              nn.Conv2d(64, 128, 3, padding=1), nn.ReLU(), nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2))
        s.v2e = nn.Linear(128 * (D // 4), E)
        s.emb = nn.Embedding(C, E)
        s.tr = nn.Transformer(E, H, NE, ND, FF, dropout=0.1, batch_first=True)
        s.out = nn.Linear(E, C)
        pe = torch.zeros(2000, E); pos = torch.arange(2000).unsqueeze(1); div = torch.exp(torch.arange(0, E, 2) * (-math.log(10000.0) / E))
        pe[:, 0::2] = torch.sin(pos * div); pe[:, 1::2] = torch.cos(pos * div)
        s.register_buffer("pe", pe)
    def forward(s, x, ys_in):
        h = s.vgg(x.unsqueeze(1))                                  # [B, 128, T/4, D/4]
        h = h.permute(0, 2, 1, 3).reshape(B, T // 4, -1)
        src = s.v2e(h) + s.pe[:T // 4]
        tgt = s.emb(ys_in) * math.sqrt(E) + s.pe[:ys_in.shape[1]]
        mask = nn.Transformer.generate_square_subsequent_mask(ys_in.shape[1], device=x.device)
        return s.out(s.tr(src, tgt, tgt_mask=mask, tgt_is_causal=True))
m = Model().cuda().to(memory_format=torch.channels_last)
opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, nesterov=True)
x = torch.randn(B, T, D, device="cuda"); ys = torch.randint(1, C - 1, (B, L), device="cuda")
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x, ys[:, :-1])
        loss = F.cross_entropy(logits.float().reshape(-1, C), ys[:, 1:].reshape(-1), label_smoothing=0.2)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    opt.step()
    return loss
for _ in range(5): step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"torch.nn modules, PyTorch-ROCm eager, bf16 autocast: {dt * 1e3:.2f} ms per inner step = {B / dt:.0f} utt/s (B = {B} x {T} frames x {D} dims)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): step()
e1.record(); torch.cuda.synchronize()
print(f"   GPU time between events: {e0.elapsed_time(e1) / n:.2f} ms per step")
