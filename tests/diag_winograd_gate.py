"""CPU numerics gate for a bf16 Winograd F(2x2, 3x3) form of conv2 / conv4 (VERDICT r5 next #1a).  No GPU, nothing of the product.

Question: if the two same-width 3x3 convs of the VGG front-end (64->64 on the full map, 128->128 on the pooled map;
mono_transformer_torch.py:49-60) ran as Winograd F(2x2, 3x3) on the MFMA -- fp32 input transform B^T d B rounded to bf16, filter
transform G g G^T rounded to bf16, the 16 element-wise channel contractions in fp32, fp32 output transform A^T m A -- forward AND
input gradient (the weight gradients stay direct), would the headline-shape step still sit inside the parity bounds of
tests/test_hip_fullsize.py (loss within 3e-4 of the reference golden; per-tensor gradient norms within 3e-2, small gradients
element-wise within 0.06 of the reference)?

The oracle (oracle/ref_cpu.py, fp32 torch on the CPU, bf16 rounding points of the engine emulated) is run three ways on the golden's
batch: (a) as it is (direct convs), (b) conv2 / conv4 through the Winograd emulation, (c) the fp32 reference values from
tests/golden/hkust_fullsize.npz.  Prints loss and per-tensor gradient distances (b)-(c) beside (a)-(c).

    python tests/diag_winograd_gate.py [--idim 80] [--B 16]      (under tests/: like every checker script it imports oracle/)
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import masr_amd  # noqa: E402,F401  (package alias)
from masr_amd.model import reference_init_state_dict  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.make_goldens import fullsize_batch  # noqa: E402

BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def q16(x):
    return x.bfloat16().float()


def winograd_conv_bf16(x, w, stats=None):
    """x [B,C,H,W] (bf16-valued fp32), w [K,C,3,3] fp32 master weights -> [B,K,H,W] fp32: F(2x2,3x3), pad 1, the operands of the 16
    contractions rounded to bf16, fp32 accumulation (one batch item at a time: 1.3 GB per transformed map at the headline shape)."""
    B_, C, H, W = x.shape
    K = w.shape[0]
    th, tw = (H + 1) // 2, (W + 1) // 2
    U = q16(torch.einsum("xi,kcij,yj->kcxy", G, w, G))                       # [K,C,4,4]
    out = torch.empty(B_, K, 2 * th, 2 * tw)
    for b in range(B_):
        xp = F.pad(x[b:b + 1], (1, 2 * tw - W + 1, 1, 2 * th - H + 1))
        d = xp.unfold(2, 4, 2).unfold(3, 4, 2)[0]                            # [C,th,tw,4,4]
        V = torch.einsum("xi,ctsij,yj->ctsxy", BT, d, BT)
        if stats is not None:
            stats.append(float(V.abs().mean() / d.abs().mean()))
        V = q16(V)
        M = torch.einsum("kcxy,ctsxy->ktsxy", U, V)                          # fp32 accumulate over the channels
        Y = torch.einsum("ix,ktsxy,jy->ktsij", AT, M, AT)                    # [K,th,tw,2,2]
        out[b] = Y.permute(0, 1, 3, 2, 4).reshape(K, 2 * th, 2 * tw)
    return out[:, :, :H, :W]


class WinoConv(torch.autograd.Function):
    """conv2d(pad 1) whose forward and input gradient run through winograd_conv_bf16; weight gradient direct (fp32 on bf16 operands)."""
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        return winograd_conv_bf16(x, w) + bias.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyq = q16(dy)                                                        # the engine keeps gradient maps in bf16
        wrot = w.flip(2, 3).transpose(0, 1).contiguous()                     # dgrad = conv with the rotated, transposed filter
        dx = winograd_conv_bf16(dyq, wrot)
        dw = torch.nn.grad.conv2d_weight(x, w.shape, dyq, padding=1)
        return dx, dw, dyq.sum((0, 2, 3))


def extract_feat_winograd(p, xs_pad, ilens):
    q = ref_cpu._q
    x = xs_pad.unsqueeze(1)
    x = q(F.relu(F.conv2d(x, p["feat_extractor.0.weight"], p["feat_extractor.0.bias"], padding=1)))
    x = q(F.relu(WinoConv.apply(x, p["feat_extractor.2.weight"], p["feat_extractor.2.bias"])))
    x = F.max_pool2d(x, 2, stride=2)
    x = q(F.relu(F.conv2d(x, q(p["feat_extractor.5.weight"]), p["feat_extractor.5.bias"], padding=1)))
    x = q(F.relu(WinoConv.apply(x, p["feat_extractor.7.weight"], p["feat_extractor.7.bias"])))
    x = F.max_pool2d(x, 2, stride=2)
    enc_lens = torch.floor(ilens.to(torch.float32) / 4).to(torch.int64)
    B_, C, Tp, Dp = x.shape
    x = x.transpose(1, 2).contiguous().view(B_, Tp, C * Dp)
    x = x @ q(p["vgg2enc.weight"]).t() + p["vgg2enc.bias"]
    return x, enc_lens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--idim", type=int, default=80)
    ap.add_argument("--B", type=int, default=16)
    args = ap.parse_args()
    HK = dict(idim=args.idim, nheads=8, d_model=512, d_inner=2048, dropout=0.0, pos_dropout=0.0, tgt_share_weight=1,
              encoder=dict(nlayers=2), decoder=dict(nlayers=4), meta={"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}})
    g = np.load(ROOT / "tests" / "golden" / "hkust_fullsize.npz")
    pre = f"d{args.idim}/"
    torch.manual_seed(531)
    sd = reference_init_state_dict(HK, 367)
    xs, il, ys, ol = fullsize_batch(args.idim)
    if args.B < 16:
        xs, il, ys, ol = xs[:args.B], il[:args.B], ys[:args.B], ol[:args.B]
    pe = ref_cpu.sinusoid_pe(3000, 512)

    # a quick unit check of the emulation itself: unrounded Winograd == direct conv
    xt, wt = torch.randn(1, 8, 10, 9), torch.randn(5, 8, 3, 3)
    global q16
    keep = q16
    q16 = lambda t: t                                                        # noqa: E731
    assert torch.allclose(winograd_conv_bf16(xt, wt), F.conv2d(xt, wt, padding=1), atol=1e-4)
    q16 = keep

    def run(wino):
        orig = ref_cpu.extract_feat
        if wino:
            ref_cpu.extract_feat = extract_feat_winograd
        try:
            with ref_cpu.bf16_emulation():
                p = ref_cpu.leafify(dict(sd, **{"pos_encoder.pe": pe}), HK)
                t0 = time.time()
                info, grads, _, _ = ref_cpu.run_batch_train(p, HK, (xs, il, ys, ol.clone()), 0.2)
                print(f"  [{'winograd' if wino else 'direct'}] {time.time() - t0:.0f} s, loss {info['loss']:.6f}", flush=True)
        finally:
            ref_cpu.extract_feat = orig
        return info, {k: v.detach().clone() for k, v in grads.items()}

    print(f"headline shape idim {args.idim}, B = {xs.shape[0]}: the bf16-emulating oracle with direct convs / with Winograd conv2 + conv4", flush=True)
    ia, ga = run(False)
    ib, gb = run(True)
    full = xs.shape[0] == 16
    if full:
        ref_loss = float(g[pre + "loss"])
        print(f"loss: reference {ref_loss:.6f}; direct {ia['loss']:.6f} (rel {abs(ia['loss'] - ref_loss) / ref_loss:.2e}); "
              f"winograd {ib['loss']:.6f} (rel {abs(ib['loss'] - ref_loss) / ref_loss:.2e})   [gate: 3e-4]")
    print(f"loss winograd vs direct: rel {abs(ia['loss'] - ib['loss']) / ia['loss']:.2e}")
    rows = []
    for n in ga:
        a, b = ga[n].double(), gb[n].double()
        rows.append((float((a - b).norm() / (a.norm() + 1e-30)), n))
    rows.sort(reverse=True)
    print("per-tensor gradient rel-L2, winograd vs direct emulation (worst 12):")
    for r, n in rows[:12]:
        print(f"  {r:.4f}  {n}")
    if full:
        worst_n = {"direct": ("", 0.0), "winograd": ("", 0.0)}
        for k in g.files:
            if not k.startswith(pre + "gradfp/"):
                continue
            n = k[len(pre + "gradfp/"):]
            if n == "pre_embed.weight" or n.endswith("in_proj_bias"):
                continue
            ref_l2 = float(g[k][2])
            for tag, gr in (("direct", ga), ("winograd", gb)):
                r = abs(float(gr[n].double().norm()) - ref_l2) / (ref_l2 + 1e-12)
                if r > worst_n[tag][1]:
                    worst_n[tag] = (n, r)
        print("worst per-tensor gradient-NORM deviation from the reference [gate 3e-2]:", worst_n)
        for k in g.files:
            if k.startswith(pre + "grad/"):
                n = k[len(pre + "grad/"):]
                b = torch.from_numpy(g[k])
                ra, rb = ga[n], gb[n]
                if n.endswith("in_proj_bias"):
                    E = 512
                    ra, rb, b = torch.cat([ra[:E], ra[2 * E:]]), torch.cat([rb[:E], rb[2 * E:]]), torch.cat([b[:E], b[2 * E:]])
                print(f"  {n}: rel-L2 vs reference: direct {float((ra - b).norm() / b.norm()):.4f}, winograd {float((rb - b).norm() / b.norm()):.4f}   [gate 0.06]")


if __name__ == "__main__":
    main()
