"""Diagnostic (not a test): which rounding owns the residual of d(loss)/d(feat_extractor.0.weight) against the oracle?
Run once as is and once with MASR_NO_FUSED_CONV1_WGRAD=1 (the unfused path contracts d(conv1 output) with the FP32 network input in
fp32 FMAs; the fused epilogue of conv3x3_resw_w1_kernel rounds both operands of that contraction to bf16 for the MFMA):

    python tests/diag_conv1_wgrad.py ; MASR_NO_FUSED_CONV1_WGRAD=1 python tests/diag_conv1_wgrad.py

Prints, for the tiny 2e2d model on the ragged golden batch and for the hkust model at the bench shape, the rel-L2 distance of the conv
gradients to the fp32 oracle, to the oracle with the engine's forward rounding points emulated, and (second process) between the two paths."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import masr_amd  # noqa: E402,F401
from masr_amd.engine import MasrEngine  # noqa: E402
from masr_amd.model import reference_init_state_dict  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.make_goldens import ODIM, TINY, fullsize_batch, synth_batch  # noqa: E402

NAMES = ["feat_extractor.0.weight", "feat_extractor.0.bias", "feat_extractor.2.weight", "feat_extractor.5.weight", "feat_extractor.7.weight", "vgg2enc.weight"]


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


def main():
    tag = "unfused (fp32 x, fp32 FMAs)" if os.environ.get("MASR_NO_FUSED_CONV1_WGRAD") else "fused (bf16 x, bf16 dY tile, MFMA)"
    out = {}
    HK = {"idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.0, "pos_dropout": 0.0, "tgt_share_weight": 1,
          "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}}
    torch.manual_seed(531)
    sd_hk = reference_init_state_dict(HK, ODIM)
    cases = [("tiny/ragged", TINY, ref_cpu.deterministic_state_dict(TINY, ODIM, seed=7), synth_batch(11, [64, 52, 40, 33], [9, 7, 5, 3])),
             ("hkust/B16xT1000", HK, sd_hk, fullsize_batch(80))]
    for name, cfg, sd, (xs, il, ys, ol) in cases:
        eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
        eng.load_state_dict(sd)
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        eng.read_stats()
        g = {k: v.cpu().clone() for k, v in eng.state_dict(flat=eng.grads).items()}
        sdp = dict(sd)
        sdp.setdefault("pos_encoder.pe", ref_cpu.sinusoid_pe(3000, cfg["d_model"]))
        p = ref_cpu.leafify(sdp, cfg)
        _, g32, _, _ = ref_cpu.run_batch_train(p, cfg, (xs, il, ys, ol.clone()), 0.2)
        with ref_cpu.bf16_emulation():
            pq = ref_cpu.leafify(sdp, cfg)
            _, gq, _, _ = ref_cpu.run_batch_train(pq, cfg, (xs, il, ys, ol.clone()), 0.2)
        print(f"--- {name}: {tag}")
        for n in NAMES:
            print(f"  {n:28s} vs fp32 oracle {rel(g[n], g32[n]):.4f}   vs emulating oracle {rel(g[n], gq[n]):.4f}   (emulating vs fp32 {rel(gq[n], g32[n]):.4f})")
            out[f"{name}/{n}"] = g[n].numpy()
    path = ROOT / "gpurun_out" / "diag_conv1_wgrad.npz"
    if path.exists() and os.environ.get("MASR_NO_FUSED_CONV1_WGRAD"):
        prev = np.load(path)
        for k in prev.files:
            if k.endswith("feat_extractor.0.weight") or k.endswith("feat_extractor.0.bias"):
                print(f"  fused vs unfused path, {k}: rel-l2 {rel(torch.from_numpy(prev[k]), torch.from_numpy(out[k])):.5f}")
    else:
        path.parent.mkdir(exist_ok=True)
        np.savez(path, **out)


if __name__ == "__main__":
    main()
